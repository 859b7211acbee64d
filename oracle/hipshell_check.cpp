// oracle/hipshell_check.cpp -- TEST INFRASTRUCTURE: instantiates include/reference_shell/HipSphTree.h for 1, 2 and 3
// dimensions against the reference's own headers, so that the binding INTEGRATION.md describes is known to compile and,
// linked with libgandalf_hip.so, to resolve every gh_* symbol it uses.  Built by `make -f oracle/ref.mk hipshell`;
// never run, never shipped.
#include "HipSphTree.h"
template class HipSphTree<1>;
template class HipSphTree<2>;
template class HipSphTree<3>;
int main() { return 0; }
