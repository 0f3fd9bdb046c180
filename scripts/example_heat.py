"""README example, runnable: heat control on a 64x64 P1 mesh, 16 time levels, one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from control_amd.fem import unit_square_p1
from control_amd.blocks import instationary_blocks
from control_amd.multiblock import ChebSpec, DirichletBCNullspace, MultiBlockSystem, SchurPC

sd, n_t, beta = unit_square_p1(64), 16, 1e-4
tau = 2.0 / (n_t - 1)
b00, b01, b10, b11, m = instationary_blocks(sd.M, sd.K, tau, beta, n_t, CN=False)
ns = tuple(DirichletBCNullspace(sd.boundary) for _ in range(m))
system = MultiBlockSystem(sd.n_dofs, sd.n_dofs, b00, b01, b10, b11, n_blocks_00=m,
                          n_blocks_11=m, nullspace_0=ns, nullspace_1=ns)
pc = SchurPC(kind="BE", M=sd.M, beta=beta, bc_nodes=sd.boundary, n_t=n_t, tau=tau,
             mass=ChebSpec(20, 0.5, 2.0),
             # sweeps replacing the reference's AMG sub-solves: degree and one interval per
             # sub-solve matrix from Lanczos estimates of their spectra on the device
             schur=ChebSpec(-1, 0.0, 0.0))
# right-hand sides: tau * M v_d for a desired state, zero force (control.py:2991-3130)
X = sd.coords
v_d = np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1])
b_0 = np.stack([tau * (sd.M @ v_d) * (i < n_t - 1) for i in range(n_t)])
b_0[:, sd.boundary] = 0.0
b_1 = np.zeros((m, sd.n_dofs))
v, zeta = np.zeros((m, sd.n_dofs)), np.zeros((m, sd.n_dofs))
ksp = system.solve(v, zeta, b_0, b_1, pc_fn=pc,
                   solver_parameters={"linear_solver": "gmres", "gmres_restart": 10,
                                      "relative_tolerance": 1e-6, "absolute_tolerance": 0.0,
                                      "maximum_iterations": 100, "monitor_convergence": False})
print(f"reason {ksp.getConvergedReason()}, {ksp.getIterationNumber()} iterations, "
      f"|v - v_d| / |v_d| at mid time = "
      f"{np.linalg.norm(v[n_t // 2] - v_d) / np.linalg.norm(v_d):.3f}")
