import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_decode import _tiny
from unsloth_amd.models.decode import DecodeEngine
model = _tiny(True); model.eval()
ids = torch.randint(0, 1000, (1, 9), generator=torch.Generator().manual_seed(13)).cuda()
eng = DecodeEngine(model, max_seq_len=128)
toks = [5, 17, 900, 33, 2]
runs = []
for r in range(3):
    lg = [eng.prefill(ids).clone()]
    for t in toks:
        lg.append(eng.step(torch.tensor([t], device="cuda")).clone())
    runs.append(lg)
for r in (1, 2):
    print("run", r, "vs run 0: max |dlogits| per step:", [float((a - b).abs().max()) for a, b in zip(runs[0], runs[r])])
g1 = torch.Generator("cuda").manual_seed(3); g2 = torch.Generator("cuda").manual_seed(3)
p = torch.softmax(runs[0][0], -1)
print("multinomial same seed:", [int(torch.multinomial(p, 1, generator=g1)) for _ in range(4)], [int(torch.multinomial(p, 1, generator=g2)) for _ in range(4)])
