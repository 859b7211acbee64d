# oracle/ref.mk -- TEST INFRASTRUCTURE, not product code.
#
# Compiles the GANDALF reference (v0.4.0) *from the sources where they lie* under
# $(REF) (default /root/reference) into oracle/_ref/ and links them with our own
# small driver oracle/ref_dump.cpp.  Nothing is copied out of the reference tree; the
# only outputs are object files, libgandalf_ref.a and the ref_dump binary, all inside
# oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).
#
# We do NOT run the reference's own Makefile.  The flags below restate its
# COMPILER_MODE=STANDARD / PRECISION=DOUBLE / OUTPUT_LEVEL=0 / DEBUG_LEVEL=1 choice
# (reference src/Makefile:65-67, 104-110, 150-170): -O3, no -ffast-math (the FAST mode
# both segfaults under g++ 11 and would void bit-level comparisons).
#
# OpenMP: the reference's "#pragma omp ... default(none)" regions do not compile with
# g++ >= 9 (const locals are no longer predetermined shared).  GCC macro-expands the
# tokens of "#pragma omp" lines, so `-Dnone=shared` turns every default(none) into
# default(shared) without touching a source file.  The only other use of the token
# `none` in the reference is the enumerator Flags.h:31, which is renamed consistently
# in every translation unit.  Set REF_OPENMP=0 for a serial build that needs no macro.
#
# Usage:  make -f oracle/ref.mk -j8            (from the repo root)

REF        ?= /root/reference
REF_OPENMP ?= 1
OUT        := oracle/_ref
CXX        ?= g++

SRCDIRS := Common GradhSph Hydrodynamics Ic MeshlessFV Nbody Radiation SM2013 Thermal Tree Feedback
# The translation units the reference's own OBJ list names (src/Makefile:176-216, non-MPI),
# plus Exception (src/Makefile:253).  Stale files the reference does not build itself
# (e.g. Ic/KhiIc.cpp) and the front ends (gandalf.cpp, Render.cpp) are left out.
UNITS := Parameters SimUnits Simulation Hydrodynamics SphSimulation Sph GradhSphSimulation \
  GradhSph SM2012SphSimulation SM2012Sph MeshlessFVSimulation FV MeshlessFV MfvCommon \
  MfvMusclSimulation MfvMuscl MfvRungeKuttaSimulation MfvRungeKutta NbodySimulation M4Kernel \
  QuinticKernel GaussianKernel TabulatedKernel Integration SphIntegration SphLeapfrogKDK \
  SphLeapfrogDKD MfvIntegration RiemannSolver SphNeighbourSearch HydroTree Tree KDTree OctTree \
  BruteForceTree MeshlessFVTree GradhSphTree SM2012SphTree Ewald AdiabaticEOS BarotropicEOS \
  Barotropic2EOS PolytropicEOS IsothermalEOS RadwsEOS LocallyIsothermal DiscLocallyIsothermal \
  RadiativeFB IonisingRadiationEOS MCRadiationEOS MultipleSourceIonisation KDRadiationTree \
  TreeMonteCarlo MonochromaticIonisationMonteCarlo TreeRay TreeRayOnTheSpot EnergyEquation \
  EnergyRadws OpacityTable Nbody NbodyLeapfrogKDK NbodyLeapfrogDKD NbodyHermite4 NbodyHermite4TS \
  NbodyHermite6TS NbodySystemTree Sinks Ghosts SphSnapshot CodeTiming Dust Particle RandomNumber \
  Supernova SupernovaDriver Ic BasicIc BinaryAccretionIc BlobIc BondiAccretionIc BossBodenheimerIc \
  ContactDiscontinuityIc DiscIc DustyBoxIc EvrardCollapseIc EwaldIc FilamentIc GaussianRingIc \
  GreshoVortexIc HierarchicalSystemIc IsothermalSphereIc KelvinHelmholtzIc NohIc PlummerSphereIc \
  PolytropeIc RayleighTaylorIc SedovBlastwaveIc ShearflowIc ShocktubeIc Shock2DIc SilccIc \
  SoundwaveIc SpitzerExpansionIc TurbulentCoreIc UniformIc ICRegularization Exception
OBJS  := $(addprefix $(OUT)/obj/,$(addsuffix .o,$(UNITS))) $(OUT)/obj/chealpix.o

CXXFLAGS := -O3 -fPIC -fno-exceptions -fno-rtti -DGANDALF_DOUBLE_PRECISION \
            -I$(REF)/src/Headers -I$(REF)/src/Common -I$(REF)/src/Ic -w
ifeq ($(REF_OPENMP),1)
CXXFLAGS += -fopenmp -Dnone=shared
endif

vpath %.cpp $(addprefix $(REF)/src/,$(SRCDIRS))
vpath %.c   $(REF)/src/Radiation

all: $(OUT)/ref_dump

$(OUT)/obj:
	mkdir -p $@

$(OUT)/obj/Exception.o: Exception.cpp | $(OUT)/obj
	$(CXX) $(CXXFLAGS) -fexceptions -c $< -o $@

$(OUT)/obj/%.o: %.cpp | $(OUT)/obj
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(OUT)/obj/chealpix.o: chealpix.c | $(OUT)/obj
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(OUT)/libgandalf_ref.a: $(OBJS)
	rm -f $@ && ar rcs $@ $(OBJS)

$(OUT)/ref_dump: oracle/ref_dump.cpp $(OUT)/libgandalf_ref.a
	$(CXX) $(CXXFLAGS) -fexceptions oracle/ref_dump.cpp -o $@ \
	    -Wl,--whole-archive $(OUT)/libgandalf_ref.a -Wl,--no-whole-archive

clean:
	rm -rf $(OUT)

.PHONY: all clean

# ---- compile + link check of the reference-side binding (include/reference_shell/HipSphTree.h) against the reference's
#      headers and libgandalf_hip.so.  Build container only; the binary is never run.
hipshell: $(OUT)/hipshell_check $(OUT)/ref_hipshell
# ... and the driver that EXECUTES it: ref_dump with the reference's GradhSphSimulation running on HipSphTree ("hipsteps" etc.;
# tests/test_gpu_boundary.py).  The run path is relative, so the binary finds libgandalf_hip.so on the GPU box too.
$(OUT)/ref_hipshell: oracle/ref_dump.cpp include/reference_shell/HipSphTree.h include/gandalf_hip.h gandalf_amd/csrc/libgandalf_hip.so $(OUT)/libgandalf_ref.a
	$(CXX) $(CXXFLAGS) -fexceptions -DREF_HIPSHELL -Iinclude -Iinclude/reference_shell oracle/ref_dump.cpp -o $@ \
	    -Wl,--whole-archive $(OUT)/libgandalf_ref.a -Wl,--no-whole-archive -Lgandalf_amd/csrc -lgandalf_hip \
	    -Wl,-rpath,'$$ORIGIN/../../gandalf_amd/csrc' -Wl,--unresolved-symbols=ignore-in-shared-libs
$(OUT)/hipshell_check: oracle/hipshell_check.cpp include/reference_shell/HipSphTree.h include/gandalf_hip.h gandalf_amd/csrc/libgandalf_hip.so | $(OUT)/obj
	$(CXX) -O0 -fno-exceptions -DGANDALF_DOUBLE_PRECISION -I$(REF)/src/Headers -I$(REF)/src/Common -Iinclude -Iinclude/reference_shell -w \
	  oracle/hipshell_check.cpp -o $@ -Lgandalf_amd/csrc -lgandalf_hip -L$(OUT) -lgandalf_ref -fopenmp -Wl,-rpath,$(abspath gandalf_amd/csrc) -Wl,--unresolved-symbols=ignore-in-shared-libs
