"""Launcher for the reference's UNCHANGED driver scripts on MI355X: replaces `horovodrun -np N python <driver> ...`
(run_scripts/pt_alpro.sh:8, run_scripts/ft_msrvtt_ret.sh) by one process per GPU under torch.distributed.run (RCCL).

    python -m alpro_amd.launch --nproc 8 --reference /path/to/ALPRO src/pretrain/run_pretrain_sparse.py \
        --config config_release/pretrain_alpro.json --output_dir /tmp/out

What it arranges before exec'ing `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` from the reference root:
  PYTHONPATH = <this repo> : <this repo>/alpro_amd/compat : <reference root>
      * `import horovod.torch`, `from apex import amp` resolve to alpro_amd/compat (torch.distributed facade, fp16=0 amp),
      * `src.modeling.alpro_models`, `src.modeling.xbert`, `src.modeling.timesformer.vit`, `src.utils.load_save` resolve to this
        repo's `src/` package, every other `src.*` module (datasets, configs, optimization, ...) to the reference's (src/__init__.py),
  MASTER_ADDR = 127.0.0.1 (single node), HSA_ENABLE_IPC_MODE_LEGACY = 0 (dmabuf IPC for RCCL),
  ALPRO_COMPUTE_DTYPE from --dtype (fp16 default: the drivers' `amp.scale_loss` supplies the loss scaling; bf16; fp32 = exact mode).
`hvd.init()` (alpro_amd.dist.init) then reads RANK / LOCAL_RANK / WORLD_SIZE from the environment torchrun sets.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_command(nproc, reference, script, script_args, port=29511, dtype="fp16", env=None):
    env = dict(os.environ if env is None else env)
    paths = [REPO, os.path.join(REPO, "alpro_amd", "compat")] + ([reference] if reference else [])
    if env.get("PYTHONPATH"):
        paths.append(env["PYTHONPATH"])
    env["PYTHONPATH"] = os.pathsep.join(paths)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ALPRO_COMPUTE_DTYPE"] = dtype
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(script_args)
    return cmd, env, (reference or os.getcwd())


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--nproc", type=int, default=8, help="processes = GPUs on this node (horovodrun -np)")
    ap.add_argument("--reference", default=os.environ.get("ALPRO_REFERENCE"), help="root of the salesforce/ALPRO checkout holding the driver")
    ap.add_argument("--port", type=int, default=29511)
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32"],
                    help="operand dtype of the GEMM / attention kernels.  fp16 (default): VTC logits within 1e-3 of the fp32 reference; the drivers' own "
                         "`amp.scale_loss` provides the loss scaling it needs.  bf16: ~2.5 %% faster, 8e-3.  fp32: exact mode (fp32 MFMA)")
    ap.add_argument("--dry-run", action="store_true", help="print the command and the environment it would run with")
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    cmd, env, cwd = build_command(a.nproc, a.reference, a.script, a.script_args, a.port, a.dtype)
    if a.dry_run:
        print("cd", cwd)
        for k in ("PYTHONPATH", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "ALPRO_COMPUTE_DTYPE"):
            print("%s=%s" % (k, env[k]))
        print(" ".join(cmd))
        return 0
    os.chdir(cwd)
    os.execvpe(cmd[0], cmd, env)


if __name__ == "__main__":
    sys.exit(main())
