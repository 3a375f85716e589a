"""Child process of test_gpu_xenc.py::test_row_major_attention_output_route: one fused cross-encoder forward in attention
mode 1 with whatever attention core the environment selects (SPR_ATTN_CORE is read once per process); the output
goes to the file named on the command line."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoints_registration_amd import ops, synthetic
from superpoints_registration_amd.transformers import TransformerCrossEncoder, TransformerCrossEncoderLayer, make_segments
dev = torch.device('cuda:0')
enc = TransformerCrossEncoder(TransformerCrossEncoderLayer(256, 8, 1024, 0.0, 'relu', True, True, True, 'dot_prod'), 2, torch.nn.LayerNorm(256))
synthetic.fill_parameters(enc, seed=1); enc = enc.to(dev)
g = torch.Generator().manual_seed(0)
lens_s, lens_t = [300, 129, 1], [257, 64, 33]
T = sum(lens_s) + sum(lens_t)
x = torch.randn(T, 256, generator=g).to(dev); pos = (torch.rand(T, 256, generator=g) * 2 - 1).to(dev)
cu, ss, sc, mx = make_segments(lens_s, lens_t, dev)
ops.set_attn_mode(1)
with torch.no_grad():
    y = enc.forward_packed(x, cu, ss, sc, mx, pos=pos, pos_bound=1.0)
torch.cuda.synchronize()
torch.save(y.cpu(), sys.argv[1])
