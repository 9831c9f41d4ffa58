import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import torch
from tango_amd import _hip as H
import bench_configs as B
H.set_stream(torch.cuda.current_stream().cuda_stream)
for nb, n in ((32768, 1024), (16384, 2048), (8192, 4096)):
    B.one_field(f'mid', nb, n, 1.0, (1,))
B.three_field()
