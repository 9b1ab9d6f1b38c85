"""Register / spill report and K-loop census of the built kernels (CPU; llvm-readelf + llvm-objdump on alpro_amd/lib/obj/<unit>.o).

    python tools/isa_report.py gemm [kernel-substring]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_object(unit, tmp):
    obj = os.path.join(ROOT, "alpro_amd", "lib", "obj", unit + ".o")
    work = os.path.join(tmp, unit + ".o")
    subprocess.run(["cp", obj, work], check=True)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], check=True, capture_output=True, cwd=tmp)
    dev = [f for f in os.listdir(tmp) if "gfx950" in f and f.startswith(unit)]
    return os.path.join(tmp, dev[0])


def main():
    unit = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        dev = device_object(unit, tmp)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev], check=True, capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s*(\S+)", line)
            if not m:
                continue
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "vgpr_spill_count":
                if pat in cur.get("name", ""):
                    dem = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                    print("%-95s vgpr %3s (spill %2s)  sgpr %3s (spill %2s)  scratch %s" % (re.sub(r"alpro::\(anonymous namespace\)::|void ", "", dem)[:95], cur.get("vgpr_count"),
                          cur.get("vgpr_spill_count"), cur.get("sgpr_count"), cur.get("sgpr_spill_count"), cur.get("private_segment_fixed_size")))
                cur = {}


if __name__ == "__main__":
    main()
