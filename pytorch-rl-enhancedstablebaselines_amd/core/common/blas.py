"""GEMM back-end selection for the MLP forward/backward (which stay on PyTorch-ROCm, north_star).

Measured on MI355X (profiles/r1_*): for this path's shapes -- batch 256..4096, width 256 -- hipBLASLt's
heuristic picks a 256x256 macro-tile kernel: a [256,256]x[256,256] GEMM becomes ONE workgroup on a 256-CU chip
(38-64 us per GEMM, 70 % of the iteration's GPU time). rocBLAS' Tensile kernels for the same shapes take a few
microseconds. `configure()` therefore prefers rocBLAS and routes `addmm` (nn.Linear with bias) away from the
hipBLASLt epilogue path. Override with CSTR_BLAS=hipblaslt|rocblas|default.
"""
import os

_configured = None


def configure(choice: str = None) -> str:
    global _configured
    choice = (choice or os.environ.get("CSTR_BLAS", "rocblas")).lower()
    if _configured == choice:
        return choice
    import torch as th

    if choice == "rocblas":
        os.environ.setdefault("DISABLE_ADDMM_CUDA_LT", "1")  # read once by ATen's addmm
        th.backends.cuda.preferred_blas_library("cublas")
    elif choice == "hipblaslt":
        th.backends.cuda.preferred_blas_library("cublaslt")
    elif choice != "default":
        raise ValueError(f"CSTR_BLAS must be rocblas, hipblaslt or default, got {choice!r}")
    _configured = choice
    return choice
