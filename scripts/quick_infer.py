import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import synthetic as S, erfnet as E
from oracle import erfnet_oracle as O
from tests.golden_util import rel_err
dev = torch.device("cuda:0")
sd = O.synthetic_state(O.erfnet_template(20))
model = E.ERFNet(20); model.load_state_dict(sd); model = model.to(dev).eval()
x = S.synthetic_images(1, 128, 256, seed=3)
with torch.no_grad():
    ref = O.erfnet(sd, x)
    y = model(x.to(dev)); print("fp32 rel err", rel_err(y.cpu(), ref))
    model.compute_dtype = torch.bfloat16
    y = model(x.to(dev)); print("bf16 rel err", rel_err(y.cpu(), ref), "label agree", (y.cpu().max(1)[1]==ref.max(1)[1]).float().mean().item())
for cd, b in ((torch.bfloat16, 16), (None, 4)):
    model.compute_dtype = cd
    xb = torch.randn(b, 3, 512, 1024, device=dev)
    with torch.no_grad():
        for _ in range(2): model.predict_labels(xb, torch.uint8)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): model.predict_labels(xb, torch.uint8)
        torch.cuda.synchronize(); dt_ = (time.time() - t0) / 5
    print(f"compute={cd} b={b}: {dt_*1e3:.2f} ms/fwd, {b/dt_:.1f} img/s")
