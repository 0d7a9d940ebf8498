"""diagnostic: per-tensor relative L2 error of one full-width product step against the CPU oracle, with each tensor's share of
its gradient set's norm -- to see which tensors carry the large RELATIVE errors (tiny, heavily cancelling sums).
usage (GPU box): python tools/diag_step_errors.py ARITH [B]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import test_fullwidth_gpu as T
from textboxgan_amd.training_step import build_trainer_state
arith = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
cfg, batch, rand, init, st, ref_losses, ref_grads = T._oracle_step(B, (False, False))
prod = build_trainer_state(cfg, dev, seed=0, compute_dtype=arith)
prod["generator"].load_state_dict({k: v.clone() for k, v in init["G"].items()})
prod["discriminator"].load_state_dict({k: v.clone() for k, v in init["D"].items()})
ts = prod["training_step"]
b = {k: v.to(dev) for k, v in batch.items()}
losses = ts.dist_train_step(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4, rand=T._todev(rand, dev))
torch.cuda.synchronize()
print("losses", [float(x) for x in losses[0] + losses[1]] + [float(losses[2])], "ref", ref_losses)
gn = [n for n in prod["generator"]._flat.names if n.startswith(("latent_encoder.", "synthesis."))]
on = [n for n in prod["generator"]._flat.names if n.startswith(("synthesis.", "word_encoder."))]
for label, names, views, ref in (("g", gn, ts.g_views, ref_grads["g"]), ("ocr", on, ts.o_views, ref_grads["ocr"]),
                                 ("d", prod["discriminator"]._flat.names, ts.d_views, ref_grads["d"])):
    tot = torch.sqrt(sum(ref[n].double().square().sum() for n in names))
    rows = []
    for n, v in zip(names, views):
        e = ref[n].double(); a = v.detach().double().cpu()
        rows.append((float((a - e).norm() / (e.norm() + 1e-30)), float(e.norm() / tot), float((a - e).norm() / tot), v.numel(), n))
    rows.sort(reverse=True)
    print(f"--- set {label}: |set| = {float(tot):.4e}; worst relative errors (rel_err, share of set norm, abs_err/|set|, numel, name)")
    for r in rows[:12]:
        print("   %.3e  %.3e  %.3e  %8d  %s" % r)
    print("   flat rel L2:", float(torch.sqrt(sum((v.detach().double().cpu() - ref[n].double()).square().sum() for n, v in zip(names, views))) / tot))
