"""Exact solution of the 1-D Riemann problem for an ideal gas (Toro, "Riemann Solvers and Numerical Methods for Fluid
Dynamics", ch. 4) - the analytical curve the reference's acceptance tests compare shock tubes with (analysis/analytical.py
`shocktube`, tests/hydro_tests/test_adsod.py).  Test helper."""
import numpy as np


def exact_riemann(x, t, rhoL, uL, pL, rhoR, uR, pR, gamma, x0=0.0):
    g = gamma
    aL, aR = np.sqrt(g*pL/rhoL), np.sqrt(g*pR/rhoR)

    def f(p, rho, pk, a):
        if p > pk:                                  # shock
            A, B = 2.0/((g + 1.0)*rho), (g - 1.0)/(g + 1.0)*pk
            return (p - pk)*np.sqrt(A/(p + B)), np.sqrt(A/(p + B))*(1.0 - 0.5*(p - pk)/(p + B))
        return 2.0*a/(g - 1.0)*((p/pk)**((g - 1.0)/(2.0*g)) - 1.0), 1.0/(rho*a)*(p/pk)**(-(g + 1.0)/(2.0*g))

    p = 0.5*(pL + pR)
    for _ in range(100):
        fl, dfl = f(p, rhoL, pL, aL)
        fr, dfr = f(p, rhoR, pR, aR)
        pn = max(p - (fl + fr + uR - uL)/(dfl + dfr), 1e-12)
        if abs(pn - p) < 1e-14*(pn + p):
            p = pn
            break
        p = pn
    fl, _ = f(p, rhoL, pL, aL)
    fr, _ = f(p, rhoR, pR, aR)
    u = 0.5*(uL + uR) + 0.5*(fr - fl)
    s = (np.asarray(x, dtype=float) - x0)/t
    rho, vel, prs = np.empty_like(s), np.empty_like(s), np.empty_like(s)
    for i, si in enumerate(s):
        if si <= u:                                 # left of the contact
            if p > pL:
                sh = uL - aL*np.sqrt((g + 1.0)/(2.0*g)*p/pL + (g - 1.0)/(2.0*g))
                if si < sh:
                    rho[i], vel[i], prs[i] = rhoL, uL, pL
                else:
                    rho[i], vel[i], prs[i] = rhoL*((p/pL + (g - 1.0)/(g + 1.0))/((g - 1.0)/(g + 1.0)*p/pL + 1.0)), u, p
            else:
                head, astar = uL - aL, aL*(p/pL)**((g - 1.0)/(2.0*g))
                tail = u - astar
                if si < head:
                    rho[i], vel[i], prs[i] = rhoL, uL, pL
                elif si > tail:
                    rho[i], vel[i], prs[i] = rhoL*(p/pL)**(1.0/g), u, p
                else:
                    c = 2.0/(g + 1.0) + (g - 1.0)/((g + 1.0)*aL)*(uL - si)
                    rho[i], vel[i], prs[i] = rhoL*c**(2.0/(g - 1.0)), 2.0/(g + 1.0)*(aL + (g - 1.0)/2.0*uL + si), pL*c**(2.0*g/(g - 1.0))
        else:
            if p > pR:
                sh = uR + aR*np.sqrt((g + 1.0)/(2.0*g)*p/pR + (g - 1.0)/(2.0*g))
                if si > sh:
                    rho[i], vel[i], prs[i] = rhoR, uR, pR
                else:
                    rho[i], vel[i], prs[i] = rhoR*((p/pR + (g - 1.0)/(g + 1.0))/((g - 1.0)/(g + 1.0)*p/pR + 1.0)), u, p
            else:
                head, astar = uR + aR, aR*(p/pR)**((g - 1.0)/(2.0*g))
                tail = u + astar
                if si > head:
                    rho[i], vel[i], prs[i] = rhoR, uR, pR
                elif si < tail:
                    rho[i], vel[i], prs[i] = rhoR*(p/pR)**(1.0/g), u, p
                else:
                    c = 2.0/(g + 1.0) - (g - 1.0)/((g + 1.0)*aR)*(uR - si)
                    rho[i], vel[i], prs[i] = rhoR*c**(2.0/(g - 1.0)), 2.0/(g + 1.0)*(-aR + (g - 1.0)/2.0*uR + si), pR*c**(2.0*g/(g - 1.0))
    return rho, vel, prs


def l1_error(x, y, y_exact):
    """L1errornorm of analysis/compute.py:109-146: mean absolute difference over the particles"""
    return float(np.sum(np.abs(np.asarray(y) - np.asarray(y_exact)))/len(x))
