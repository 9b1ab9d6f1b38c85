"""Census of the 8-phase GEMM's K loop and of the ticket register in a built gemm.o (CPU):  python tools/isa_kloop.py [objdir]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(obj):
    with tempfile.TemporaryDirectory() as tmp:
        work = os.path.join(tmp, "gemm.o")
        subprocess.run(["cp", obj, work], check=True)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], check=True, capture_output=True, cwd=tmp)
        dev = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, dev)], check=True, capture_output=True, text=True).stdout


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "alpro_amd", "lib", "obj", "gemm.o")
    dis = disassemble(obj)
    for fn in re.split(r"\n(?=[0-9a-f]{16} <)", dis):
        head = fn.split("\n", 1)[0]
        if "gemm_nt256q_kernel" not in head:
            continue
        lines = [l.split("//")[0].strip() for l in fn.split("\n")[1:]]
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        span = lines[mf[0]:mf[-1] + 1]
        cnt = lambda pat, ls: sum(1 for l in ls if re.search(pat, l))   # noqa: E731
        print(head[18:90], "lines", len(lines), "mfma", len(mf), "K-span", mf[0], mf[-1])
        print("   in K-span: lds-dma %d  vmcnt waits %d  barriers %d  scratch %d  lane-spill ops %d  atomics %d  s_load %d" % (
            cnt(r"global_load_lds", span), cnt(r"s_waitcnt.*vmcnt", span), cnt(r"\bs_barrier\b", span), cnt(r"\bscratch_", span),
            cnt(r"v_(read|write)lane", span), cnt(r"global_atomic", span), cnt(r"\bs_load_", span)))
        print("   whole kernel: scratch %d  lane-spill ops %d" % (cnt(r"\bscratch_", lines), cnt(r"v_(read|write)lane", lines)))
        for i, l in enumerate(lines):
            if re.search(r"global_atomic_(add|or)\b", l) and "sc0" in l:
                reg = l.split()[1].rstrip(",")
                uses = [(j, lines[j]) for j in range(len(lines)) if re.search(r"\b%s\b" % reg, lines[j]) and j != i]
                print("   ticket atomic at %d -> %s; other instructions naming it: %s" % (i, reg, [(j, u[:60]) for j, u in uses][:8]))


if __name__ == "__main__":
    main()
