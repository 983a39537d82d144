"""scasml_gp_amd -- MI355X (gfx950) native hot path of SCaSML_GP.

Drop-in for the reference's call surface on the path BASELINE.json names
(SURVEY.md section 8(b)):

    scasml_gp_amd.equations.equations.{Equation, Grad_Dependent_Nonlinear}
    scasml_gp_amd.solvers.MLP.MLP, .ScaSML.ScaSML,
    scasml_gp_amd.solvers.MLP_full_history.MLP_full_history,
    scasml_gp_amd.solvers.ScaSML_full_history.ScaSML_full_history
    scasml_gp_amd.models.GP.{GP, GP_Grad_Dependent_Nonlinear}

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) calling
hand-written HIP kernels through the C ABI of include/scasml_hip.h.  There is no CPU
fallback: without libscasml_hip.so or without a GPU the solvers raise.
"""
__version__ = "0.1.0"
