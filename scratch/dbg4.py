import sys, os, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'ml-pointconvformer_amd')
from conftest import load_golden, split
import pcf_layers, pcf_cuda, pcf_fused
g = load_golden('pointconv_single')
dev = torch.device('cuda:0')
class Cfg(dict):
    __getattr__ = dict.__getitem__
c = Cfg(attention_type='subtraction', BATCH_NORM=False, drop_path_rate=0., dropout_rate=0., USE_VI=False, USE_PE=False, PCONV_OPT=False, USE_CUDA_KERNEL=True, layer_norm_guidance=False)
a = split(g, 'in.')
def run(fused):
    orig = pcf_fused.rowlin_supported
    if not fused: pcf_fused.rowlin_supported = lambda *x: False
    layer = pcf_layers.PointConv(3, 32, c, weightnet=[3, 16])
    layer.load_state_dict(split(g, 'sd.'), strict=True); layer.to(dev).train()
    feats = a['dense_feats'].to(dev).requires_grad_(True)
    caps = {}
    def hk(m, i, o):
        caps['y'] = o; o.register_hook(lambda gr: caps.__setitem__('gy', gr.clone()))
    layer.linear.register_forward_hook(hk)
    out, wn = layer(a['dense_xyz'].to(dev), feats, a['nei_inds'].to(dev))
    out0 = out.detach().clone()
    gup = g['gup'].to(dev)
    gr = torch.autograd.grad(out, [feats, layer.linear.bias], gup)
    torch.cuda.synchronize()
    pcf_fused.rowlin_supported = orig
    return dict(out0=out0.cpu(), out1=out.detach().cpu(), y=caps['y'].detach().cpu(), gy=caps['gy'].cpu(), gup=gup.cpu(), gb=gr[1].cpu())
A = run(False); B = run(True)
for k in A: print(k, (A[k] - B[k]).abs().max().item())
print('out0 vs out1 fused', (B['out0'] - B['out1']).abs().max().item())
man = (B['gup'] * (B['out1'] > 0)).sum((0, 1))
print('manual bias grad vs autograd (fused)', (man - B['gb']).abs().max().item())
d = (A['gy'] - B['gy']).abs().sum(-1)[0]
print('rows with gy diff', (d > 1e-6).nonzero().flatten()[:20].tolist(), 'count', int((d > 1e-6).sum()))
