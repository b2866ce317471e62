"""Library-kernel selection for the parts of the step that stay on vendor libraries
(projection / FFN GEMMs on rocBLAS + hipBLASLt, discriminator convolutions on MIOpen).

These are not hot-path kernels ("stock torch (hipBLASLt / MIOpen) in the
build"), but which library kernel runs matters: PyTorch's TunableOp picks, per GEMM shape, the
fastest of all rocBLAS and hipBLASLt solutions (at the ViT-VQGAN batch-32 shapes the default
heuristic runs the fp32 GEMMs at ~88 TFLOP/s, the tuned choice at 120-140 TFLOP/s).  A results
file for the benchmark shapes ships next to this module so a default run does not re-tune;
shapes that are not in the file are tuned on first use.
"""
import os
import shutil
import tempfile

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
VITVQGAN_B32 = os.path.join(_HERE, "tunableop_gfx950_vitvqgan_b32.csv")


def enable_gemm_tuning(results_csv=VITVQGAN_B32, tune_missing=True, max_tuning_ms=30):
    """Turn TunableOp on for this process.  `results_csv` is used if it exists and its validators
    (PyTorch / ROCm / hipBLASLt / rocBLAS versions, gfx arch) match.  TunableOp rewrites its file
    at exit, so every process works on a private copy (ranks never share or modify the shipped one)."""
    tn = torch.cuda.tunable
    tn.enable(True)
    tn.tuning_enable(bool(tune_missing))
    tn.set_max_tuning_duration(int(max_tuning_ms))
    private = os.path.join(tempfile.mkdtemp(prefix="amk_tunableop_"), "results.csv")
    if results_csv and os.path.exists(results_csv):
        shutil.copyfile(results_csv, private)
    tn.set_filename(private, insert_device_ordinal=False)
    return private


def enable_conv_autotune(flag=True, shipped_db=True):
    """MIOpen find mode for the PatchGAN discriminator convolutions (+10 % on the whole step).
    With `shipped_db` the find / perf databases recorded for the benchmark's convolution shapes on
    gfx950 are offered to MIOpen through a private copy of MIOPEN_USER_DB_PATH, so the ~100 s of
    first-run auto-tuning are not repeated (MIOpen ignores them if its version differs).  Call
    before the first convolution runs."""
    torch.backends.cudnn.benchmark = bool(flag)
    src = os.path.join(_HERE, "miopen_gfx950")
    if flag and shipped_db and "MIOPEN_USER_DB_PATH" not in os.environ and os.path.isdir(src):
        dst = tempfile.mkdtemp(prefix="amk_miopen_")
        for name in os.listdir(src):
            shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
        os.environ["MIOPEN_USER_DB_PATH"] = dst
