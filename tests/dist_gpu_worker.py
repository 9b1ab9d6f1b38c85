"""Worker of tests/test_dist_gpu.py: one data-parallel rank (or the single-process reference) of a VTC training step on the HIP
path.  Usage: python tests/dist_gpu_worker.py <out.pt> <B per rank> <wire: fp32|bf16>   (RANK / WORLD_SIZE from the environment)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    out_path, B, wire = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    from alpro_amd import config as rt, dist, hip
    from alpro_amd.modeling.alpro_models import AlproForVideoTextRetrieval
    from alpro_amd.optim import FlatAdamW
    from tests.golden.det_init import det_batch, fill_state_dict_
    from tests.conftest import BERT_CFG
    from tests.test_host_cpu import VENC, make_cfg
    dist.init()
    world, rank = dist.size(), dist.rank()
    torch.cuda.set_device(0)
    hip.load()
    rt.set_compute_dtype("fp32")
    dist.set_allgather_grad_mode("sum")   # exact full-batch gradient: N ranks x B == 1 rank x N*B
    T = 2
    m = AlproForVideoTextRetrieval(make_cfg(dict(BERT_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)), dict(VENC, num_frm=T, drop_path_rate=0.0))
    fill_state_dict_(m)
    m.cuda().train()
    dist.broadcast_parameters(m)            # (identity values on every rank; under ALPRO_FORCE_COLLECTIVES it exercises the broadcast path)
    total = 2 * B                           # the global batch is the same whether 2 ranks x B or 1 rank x 2B run it
    full = det_batch(total, T, seed_name="dist_gpu", with_mlm=False, with_mpm=False)
    per = total // world
    mine = {k: (v[rank * per:(rank + 1) * per].cuda() if torch.is_tensor(v) else v) for k, v in full.items()}
    opt = FlatAdamW(m.parameters(), lr=0.0, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.0, max_grad_norm=None, overlap_backward=True,
                    wire_dtype=torch.bfloat16 if wire == "bf16" else None)

    def vtc_only(batch):
        ve = m._forward_visual_embeds(batch["visual_inputs"])
        te = m._text_embeds(batch["text_input_ids"], batch["text_input_mask"])
        return m._vtc(m._video_feat(ve), m._text_feat(te))[0]

    losses = []
    for it in range(2):                     # step 0 builds the flat buffers (lr = 0: parameters do not move); step 1 runs the overlapped exchange
        loss = vtc_only(mine)
        loss.backward()
        losses.append(loss.detach().clone())
        if it == 0:
            opt.step()
            opt.zero_grad()
    on_wire_early = len(opt._reduced)       # ranges already handed to the all-reduce when backward returned
    opt.synchronize(average=True)
    torch.cuda.synchronize()
    names = [n for n, p in m.named_parameters() if p.grad is not None]
    pd = dict(m.named_parameters())
    keep = ["temp", "vision_proj.weight", "text_proj.bias", "visual_encoder.model.blocks.11.attn.qkv.bias", "visual_encoder.model.blocks.0.temporal_fc.weight",
            "visual_encoder.model.blocks.5.mlp.fc1.weight", "visual_encoder.model.pos_embed", "visual_encoder.model.norm.weight",
            "text_encoder.bert.encoder.layer.3.attention.self.value.weight", "text_encoder.bert.embeddings.LayerNorm.bias"]
    lsum = losses[1].clone()
    if dist.collectives_active():
        torch.distributed.all_reduce(lsum)
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"
    print("dist_backend: %s  world %d  collectives_active %s  ranges on the wire before backward returned: %d" % (backend, world, dist.collectives_active(), on_wire_early), flush=True)
    res = dict(world=world, rank=rank, backend=backend, loss=float(lsum / world), on_wire_early=on_wire_early, names=names,
               norms=torch.tensor([float(pd[n].grad.norm()) for n in names], dtype=torch.float64),
               grads={n: pd[n].grad.detach().float().cpu() for n in keep})
    torch.save(res, out_path)
    dist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
