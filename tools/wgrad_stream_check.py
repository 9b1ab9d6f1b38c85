"""GPU: the fp16 pretrain step with the weight gradients on a side stream (ALPRO_WGRAD_STREAM=1) against the launch-stream form.  Two processes run
the same two training steps (bench.py's model, batch and optimizer at B pairs) from the same seed; the losses and every parameter gradient of both steps
must agree bit for bit (the side stream changes when a weight-gradient kernel runs, not what it adds up).  Likewise the 2B-caption text-encoder pass on its
own side stream (ALPRO_TEXT_STREAM=1), alone and together with the weight-gradient stream.  python tools/wgrad_stream_check.py [B ...]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def worker(out, B):
    import bench
    from alpro_amd import config as rt, hip
    from alpro_amd.modeling.alpro_models import AlproForPretrain
    from alpro_amd.optim import FlatAdamW
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip.load()
    rt.set_compute_dtype("fp16")
    rt.set_cls_precise("auto")
    torch.manual_seed(1234)
    T = 8
    cfg = bench.Cfg(dict(bench.BERT_CFG, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1))
    model = AlproForPretrain(cfg, dict(bench.VENC, num_frm=T)).to(dev)
    model.train()
    opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.0, max_grad_norm=20.0)
    batch = bench.synth_batch(B, T, dev, seed=0, full=True)
    rec = {"loss": [], "sums": [], "names": [n for n, p in model.named_parameters() if p.requires_grad]}
    with torch.enable_grad():
        for it in range(2):
            o = model(batch)
            loss = o["mlm_loss"] + o["itm_loss"] + o["itc_loss"] + o["mpm_loss"]
            opt.backward(loss)
            torch.cuda.synchronize()
            rec["loss"].append(float(loss))
            pd = dict(model.named_parameters())
            # two order-sensitive fingerprints per gradient: the fp64 sum and the fp64 sum of squares of the fp32 values
            rec["sums"].append(torch.stack([torch.stack([pd[n].grad.double().sum(), pd[n].grad.double().pow(2).sum()]) if pd[n].grad is not None
                                            else torch.zeros(2, dtype=torch.float64, device=dev) for n in rec["names"]]).cpu())
            opt.step()
            opt.zero_grad()
    rec["side"] = rt.wgrad_stream_enabled()
    rec["peak_gb"] = torch.cuda.max_memory_allocated() / 2**30
    torch.save(rec, out)


def run(B, side, out, text=False, prompter=False):
    env = dict(os.environ, ALPRO_WGRAD_STREAM="1" if side else "0", ALPRO_TEXT_STREAM="1" if text else "0", ALPRO_PROMPTER_STREAM="1" if prompter else "0")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", out, str(B)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:]
    return torch.load(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    bad = 0
    for B in [int(x) for x in sys.argv[1:]] or [4, 32]:
        with tempfile.TemporaryDirectory() as d:
            a, b, c = run(B, False, os.path.join(d, "a.pt")), run(B, True, os.path.join(d, "b.pt")), run(B, False, os.path.join(d, "c.pt"))
            t, tb = run(B, False, os.path.join(d, "t.pt"), text=True), run(B, True, os.path.join(d, "tb.pt"), text=True)
            pr = run(B, True, os.path.join(d, "pr.pt"), text=True, prompter=True)
        assert not a["side"] and b["side"]
        for name, x in (("side stream vs launch stream", b), ("launch stream again (the control)", c), ("text pass on its side stream (ALPRO_TEXT_STREAM=1)", t),
                        ("text pass and weight gradients on side streams", tb), ("... and the prompter's pass on its side stream (ALPRO_PROMPTER_STREAM=1)", pr)):
            diff = [(it, n) for it in range(2) for i, n in enumerate(a["names"]) if not torch.equal(a["sums"][it][i], x["sums"][it][i])]
            same = a["loss"] == x["loss"] and not diff
            print("B = %d, %s: losses %s / %s, %d gradients x 2 steps, %d differ %s -> %s (peak %.1f / %.1f GB)"
                  % (B, name, a["loss"], x["loss"], len(a["names"]), len(diff), diff[:4], "BITWISE EQUAL" if same else "DIFFERENT", a["peak_gb"], x["peak_gb"]))
            bad += (not same) and x is not c
    sys.exit(1 if bad else 0)
