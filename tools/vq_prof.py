import os, sys
sys.path.insert(0, "/root/repo")
import torch, showo_amd
vq = showo_amd.MAGVITv2(max_batch=25, max_res=256, precision=1).cuda().eval()
x = torch.rand(25, 3, 256, 256, device="cuda") * 2 - 1
vq.get_code(x); vq.get_code(x); torch.cuda.synchronize()
