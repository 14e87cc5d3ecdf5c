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
    out, wn = layer(a['dense_xyz'].to(dev), feats, a['nei_inds'].to(dev))
    torch.cuda.synchronize()
    out.backward(g['gup'].to(dev))
    torch.cuda.synchronize()
    pcf_fused.rowlin_supported = orig
    return {n: p.grad.cpu() for n, p in layer.named_parameters()}, feats.grad.cpu()
ga, fa = run(False)
gb, fb = run(True)
print('torch path vs golden', (fa - g['gin.dense_feats']).abs().max().item())
print('fused path vs golden', (fb - g['gin.dense_feats']).abs().max().item())
print('linear.bias golden', g['gsd.linear.bias'][:6])
print('linear.bias torch ', ga['linear.bias'][:6])
print('linear.bias fused ', gb['linear.bias'][:6])
