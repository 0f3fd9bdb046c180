import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, common
np.set_printoptions(linewidth=200, precision=3)
for CN in (False,):
  for ksp in ("gmres","fgmres"):
    p = common.heat_problem(n=10, n_t=10, CN=CN, beta=1e-2)
    osys, gsys = common.oracle_system(p), common.gpu_system(p)
    opc = common.oracle_pc(p, (20,.5,2.), (12,.08,2.1)); gpc = common.gpu_pc(p, (20,.5,2.), (12,.08,2.1))
    m, nx = p["m"], p["sd"].n_dofs
    X = p["sd"].coords
    xs = np.stack([np.sin(np.pi*X[:,0])*np.sin(np.pi*X[:,1])*(1+0.1*k) for k in range(2*m)])
    b = osys.mult(xs.ravel()).reshape(2*m, nx)
    sp = {"linear_solver": ksp, "gmres_restart": 10, "maximum_iterations": 60, "relative_tolerance": 1e-6, "absolute_tolerance": 0.0, "monitor_convergence": False, "preconditioner": True}
    uo0, uo1 = np.zeros((m, nx)), np.zeros((m, nx))
    ro = osys.solve(uo0, uo1, b[:m], b[m:], solver_parameters=sp, pc_fn=opc)
    ug0, ug1 = np.zeros((m, nx)), np.zeros((m, nx))
    rg = gsys.solve(ug0, ug1, b[:m].copy(), b[m:].copy(), solver_parameters=sp, pc_fn=gpc)
    ho, hg = np.asarray(ro.history), np.asarray(rg.history)
    print(ksp, ro.its, rg.its)
    print('ho', ho); print('abs', np.abs(hg-ho)); print('rel', np.abs(hg-ho)/ho)
    print('sol diff', common.rel_err(np.vstack([ug0,ug1]), np.vstack([uo0,uo1])), 'err vs true', common.rel_err(np.vstack([uo0,uo1]), xs), common.rel_err(np.vstack([ug0,ug1]), xs))
