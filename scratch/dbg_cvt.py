import torch, numpy as np
M = 0.4244384765625
x = torch.tensor([M, 0.530548095703125], dtype=torch.float32, device="cuda")
print("cvt tie:", hex(x[:1].half().cpu().numpy().view(np.uint16)[0]))
p = x[1:] * torch.tensor([0.8], dtype=torch.float32, device="cuda")
print("prod:", p.cpu().numpy()[0].hex() if hasattr(p.cpu().numpy()[0], "hex") else float(p), float(p) == M, hex(p.half().cpu().numpy().view(np.uint16)[0]))
