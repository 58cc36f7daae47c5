"""7B sanity check of the LoRA gradients: directional derivative. loss(B - eps*g/|g|) - loss(B) ≈ -eps*|g| (first order)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridgelang_amd import ops
from bridgelang_amd.training.lora import LoraAdapters
from bridgelang_amd.training.step import TrainStep
from bridgelang_amd.weights import allocate, openvla_7b_dims, tiny_dims
dev = torch.device("cuda:0")
dims = tiny_dims() if "--tiny" in sys.argv else openvla_7b_dims()
w = allocate(dims, dev).fill_synthetic(seed=0)
lora = LoraAdapters(w, r=32)
B, L = (16, 32) if "--tiny" not in sys.argv else (4, 24)
ts = TrainStep(w, "lora", B, L, lora=lora, max_grad_norm=float("inf"), weight_decay=0.01)
g = torch.Generator().manual_seed(0)
ids = torch.randint(3, 31000, (B, L), generator=g); ids[:, 0] = 1
ids[:, -8:-1] = torch.randint(31744, 32000, (B, 7), generator=g); ids[:, -1] = 2
labels = torch.full((B, L), -100); labels[:, -8:] = ids[:, -8:]
pv = torch.randn(B, 6, 224, 224, generator=g).to(torch.bfloat16)
ts.set_batch(ids, None, pv, labels)
loss0 = ts.forward().item()
ts.backward()
st = ts.store
gA = torch.cat([st.grad_view(f"lora.{i}.A").flatten() for i in range(len(lora.adapters))])
gB = torch.cat([st.grad_view(f"lora.{i}.B").flatten() for i in range(len(lora.adapters))])
nA, nB = gA.norm().item(), gB.norm().item()
print(f"loss0 {loss0:.5f}  |gA| {nA:.4g} (must be 0 at B = 0)  |gB| {nB:.4g}  |gB|_1 {gB.abs().sum().item():.4g}")
for target in (0.1, 0.4):
    eps = target / nB
    for i, ad in enumerate(lora.adapters):
        ad.B.copy_((-(eps / nB) * st.grad_view(f"lora.{i}.B").view(ad.B.shape)).to(torch.bfloat16))
    ops.run_all(ts._adapter_ops)
    loss1 = ts.forward().item()
    print(f"predicted dloss {-target:.4f}   measured {loss1 - loss0:.4f}")
