"""Exploration aid (not a test): loss trajectory of the 7B vla-full-train step on one repeated batch for a few learning rates."""
import gc, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_full_size_train_gpu import _dummy_batches
from bridgelang_amd.training.step import TrainStep
from bridgelang_amd.weights import allocate, openvla_7b_dims
dev = torch.device("cuda:0")
B, L = 32, 40
(ids, labels, pv), = _dummy_batches(B, L, 1, seed=11)
w = allocate(openvla_7b_dims(), dev)
for lr in [float(a) for a in sys.argv[1:]]:
    w.fill_synthetic(seed=0)
    ts = TrainStep(w, "vla-full-train", B, L, max_grad_norm=1.0, weight_decay=0.0)
    ts.set_batch(ids, None, pv, labels)
    log = []
    for _ in range(8):
        loss, norm = ts.step(lr)
        log.append((round(loss.item(), 3), round(norm.item(), 1)))
    print(lr, log, flush=True)
    del ts; gc.collect(); torch.cuda.empty_cache()
