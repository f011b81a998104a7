import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet as E, autograd as ag
from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
m = E.ERFNet(20).cuda().train()
tr = Trainer(m, CITYSCAPES_WEIGHTS)
x = torch.randn(2, 3, 64, 128, device="cuda"); t = torch.randint(0, 20, (2, 1, 64, 128), device="cuda")
n0 = ag.FUSED_BN_BACKWARD_CALLS[0]
tr.step(x, t)
torch.cuda.synchronize()
print("fused calls per step:", ag.FUSED_BN_BACKWARD_CALLS[0] - n0)
