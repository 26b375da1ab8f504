"""TEST INFRASTRUCTURE (oracle/): stages the reference's own kernel modules for a run on the GPU box.

The GPU box has no /root/reference. `python oracle/stage_reference.py` (called by `__graft_entry__.build()`
whenever /root/reference is present, i.e. in the build container) copies the handful of reference modules
the Triton kernels need into `oracle/_ref/unsloth/` -- git-ignored (never in history), NOT gpurun-ignored
(travels to the box with the snapshot, like our own built .so). `oracle/make_golden_bf16_gpu.py` imports
them there through the stub harness of SURVEY.md section 10 and runs them NATIVELY (triton-rocm, bf16) to
produce `tests/golden/ref_triton_bf16.pt`: the bf16 pin that the CPU interpreter cannot give (numpy has no
bf16).

Nothing under unsloth_amd/ reads oracle/_ref; it is only ever the checker.
"""
import hashlib
import json
import os
import shutil
import sys

REF_ROOT = os.environ.get("UNSLOTH_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "_ref", "unsloth")

# exactly the modules the stub harness ends up importing (probed: sys.modules after load_reference())
FILES = [
    "bnb_availability.py",
    "device_type.py",
    "kernels/utils.py",
    "kernels/fp8.py",
    "kernels/rms_layernorm.py",
    "kernels/rope_embedding.py",
    "kernels/swiglu.py",
    "kernels/geglu.py",
    "kernels/cross_entropy_loss.py",
    "kernels/fast_lora.py",
]


def stage(verbose=True):
    src_root = os.path.join(REF_ROOT, "unsloth")
    if not os.path.isdir(src_root):
        if verbose:
            print(f"[stage_reference] {src_root} absent: nothing staged (GPU box uses the prebuilt oracle/_ref)")
        return False
    manifest = {}
    for rel in FILES:
        src = os.path.join(src_root, rel)
        dst = os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        with open(src, "rb") as f:
            manifest[rel] = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(HERE, "_ref", "MANIFEST.json"), "w") as f:
        json.dump({"source": src_root, "sha256": manifest}, f, indent=1, sort_keys=True)
    if verbose:
        print(f"[stage_reference] staged {len(FILES)} reference modules under {DST}")
    return True


if __name__ == "__main__":
    sys.exit(0 if stage() else 1)
