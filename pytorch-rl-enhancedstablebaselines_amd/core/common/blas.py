"""GEMM back-end selection for the GEMMs that PyTorch-ROCm still evaluates: every GEMM of the learners with CSTR_FUSED_LINEAR=0
(the north_star-literal configuration, `mlp_on_pytorch_rocm_variant` in bench.py) and of the stock-ATen fallback, and the
shapes outside the hand-written f32-MFMA Linear kernels' range (more than 1024 rows with a wide input) otherwise -- the default
SAC / TD3 iteration issues no rocBLAS launch any more (profiles/r02_bench_sac_kernel_stats.csv).

Measured on MI355X (profiles/r01_bench_eager_hipblaslt_kernel_stats.csv): for this path's shapes -- batch 256..4096, width 256 -- hipBLASLt's
heuristic picks a 256x256 macro-tile kernel: a [256,256]x[256,256] GEMM becomes ONE workgroup on a 256-CU chip
(38-64 us per GEMM, 70 % of the iteration's GPU time). rocBLAS' Tensile kernels for the same shapes take a few
microseconds. `configure()` therefore prefers rocBLAS and routes `addmm` (nn.Linear with bias) away from the
hipBLASLt epilogue path. Override with CSTR_BLAS=hipblaslt|rocblas|default.

rocBLAS' own heuristic is not the fastest solution for these shapes either (e.g. the [256,256]^T x [256,256] weight
gradient: 9.3 us by default, 3.2 us with another Tensile solution). `tunableop_gfx950.csv` holds, for every GEMM shape the
SAC / TD3 / MADDPG learners issue at their class defaults, the rocBLAS solution that is fastest as a graph-replayed launch on
MI355X (produced by tools/tune_gemms.py, PyTorch TunableOp format). `configure()` loads it with tuning DISABLED: a pure
table lookup, no run-time tuning; shapes that are not in the table use the default. CSTR_TUNABLEOP_FILE=<path> selects
another table, CSTR_TUNABLEOP_FILE=0 turns the lookup off.
"""
import os

_configured = None
TUNED_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def _load_tuned_table(th) -> None:
    path = os.environ.get("CSTR_TUNABLEOP_FILE", TUNED_TABLE)
    if path == "0" or not os.path.exists(path) or os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
        return  # an explicit PYTORCH_TUNABLEOP_* environment wins
    if th.cuda.tunable.is_enabled():
        return  # the caller drives TunableOp itself (e.g. bench.py --tunable 1)
    th.cuda.tunable.enable(True)
    th.cuda.tunable.tuning_enable(False)
    th.cuda.tunable.set_filename(path)


def configure(choice: str = None) -> str:
    global _configured
    choice = (choice or os.environ.get("CSTR_BLAS", "rocblas")).lower()
    if _configured == choice:
        return choice
    import torch as th

    if choice == "rocblas":
        os.environ.setdefault("DISABLE_ADDMM_CUDA_LT", "1")  # read once by ATen's addmm
        th.backends.cuda.preferred_blas_library("cublas")
        _load_tuned_table(th)
    elif choice == "hipblaslt":
        th.backends.cuda.preferred_blas_library("cublaslt")
    elif choice != "default":
        raise ValueError(f"CSTR_BLAS must be rocblas, hipblaslt or default, got {choice!r}")
    _configured = choice
    return choice
