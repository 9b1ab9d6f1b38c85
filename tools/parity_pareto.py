"""Assemble profiles/r4_parity_pareto.txt from one gpurun call's outputs (tools/r4_final_evidence.sh):
    python tools/parity_pareto.py gpurun_out/r4 > profiles/r4_parity_pareto.txt
mode -> worst-of-four-fixture VTC-logit error (tests/test_model_parity.py::test_vtc_logits_meet_the_north_star_bar_on_every_fixture, -s output),
B = 64 proxy error against the exact fp32 HIP mode (::test_full_size_pretrain_forward_in_the_bench_dtype_vs_the_exact_mode) and ms / step of the
default benchmark in that mode (bench.py --dtype .. --cls-precise ..), all from ONE box."""
import json
import os
import re
import sys

d = sys.argv[1]
txt = open(os.path.join(d, "t_all.txt")).read()
modes = ["fp32", "fp16", "fp16_plain", "bf16", "bf16_cls"]
names = {"fp32": "fp32 exact (fp32 MFMA)", "fp16": "fp16 + precise CLS rows (bench default)", "fp16_plain": "fp16 plain (round-3 default)",
         "bf16": "bf16 plain", "bf16_cls": "bf16 + precise CLS rows"}
fix = {}
for m in re.finditer(r"\[vtc-logit parity (\S+)\] (.*)", txt):
    vals = dict(re.findall(r"(\w+) ([0-9.e+-]+)", m.group(2)))
    fix[m.group(1)] = {k: float(v) for k, v in vals.items()}
proxy = {}
for m in re.finditer(r"\[B=64 proxy (\S+)\s*\] VTC logits \(4096\): max (\S+) p99.9 (\S+) rms (\S+) \| ITM (\S+) \| MPM (\S+) \| MLM (\S+)", txt):
    proxy[m.group(1)] = [float(x) for x in m.groups()[1:]]
ms = {}
for key, f in (("fp16", "bench_pretrain_step_B64.json"), ("fp16_plain", "bench_mode_fp16_cls0.json"), ("bf16", "bench_mode_bf16_cls0.json"),
               ("bf16_cls", "bench_mode_bf16_cls1.json"), ("fp32", "bench_mode_fp32_cls0.json")):
    p = os.path.join(d, f)
    if os.path.exists(p) and os.path.getsize(p):
        j = json.loads(open(p).read().strip().splitlines()[-1])
        ms[key] = (j["ms_per_step"], j["value"])
print("# Round 4 parity Pareto table (VERDICT r3 item 1c): one MI355X box, one gpurun call, final binary.")
print("# VTC-logit error = max |logit - reference logit| against the reference-generated fixtures (tests/golden); the north-star bar is 1e-3.")
print("# B=64 proxy = AlproForPretrain eval B=64 x 8f, mode vs the exact fp32 HIP mode (itself <= 5e-6 from the reference), all 4096 VTC logits.")
print("# step = bench.py default workload (pretrain_step B=64) in that mode, deterministic reductions on.")
print()
cases = list(fix)
print("%-42s | %-44s | worst     | meets 1e-3 | B=64 proxy max / p99.9 / rms      | ITM      MLM      | ms/step  pairs/s" % ("mode", " / ".join(c.replace("pretrain_", "p_").replace("retrieval_", "r_") for c in cases)))
for k in modes:
    errs = [fix[c].get(k, float("nan")) for c in cases]
    worst = max(errs)
    px = proxy.get(k)
    print("%-42s | %-44s | %.2e  | %-10s | %-33s | %-17s | %s" % (
        names[k], " / ".join("%.1e" % e for e in errs), worst, "yes" if worst <= 1e-3 else "NO",
        ("%.2e / %.2e / %.2e" % tuple(px[:3])) if px else "(reference of the proxy)" if k == "fp32" else "-",
        ("%.1e  %.1e" % (px[3], px[5])) if px else "-",
        ("%7.2f  %7.1f" % ms[k]) if k in ms else "-"))
print()
print("# bench default = the fastest mode whose worst fixture is inside the bar: fp16 + precise CLS rows.")
