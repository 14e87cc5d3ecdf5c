import sys, os, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'ml-pointconvformer_amd')
from conftest import load_golden, split
import pcf_layers, pcf_cuda, pcf_fused
from oracle import pcf_oracle as O
g = load_golden('pointconv_single')
dev = torch.device('cuda:0')
class Cfg(dict):
    __getattr__ = dict.__getitem__
c = Cfg(attention_type='subtraction', BATCH_NORM=False, drop_path_rate=0., dropout_rate=0., USE_VI=False, USE_PE=False, PCONV_OPT=False, USE_CUDA_KERNEL=True, layer_norm_guidance=False)
layer = pcf_layers.PointConv(3, 32, c, weightnet=[3, 16])
layer.load_state_dict(split(g, 'sd.'), strict=True); layer.to(dev).train()
a = split(g, 'in.')
feats = a['dense_feats'].to(dev).requires_grad_(True)
caps = {}
def hook(m, i, o):
    o.retain_grad(); caps['w'] = o
layer.weightnet.register_forward_hook(hook)
out, wn = layer(a['dense_xyz'].to(dev), feats, a['nei_inds'].to(dev))
print('out', (out.cpu() - g['out.new_feat']).abs().max().item())
print('w', (caps['w'].cpu() - g['cap.w']).abs().max().item())
out.backward(g['gup'].to(dev))
print('gw', (caps['w'].grad.cpu() - g['gcap.w']).abs().max().item())
print('gfeats', (feats.grad.cpu() - g['gin.dense_feats']).abs().max().item())
# direct op call with golden inputs
d = lambda t: t.contiguous().to(dev)
add = torch.zeros(1, 256, 16, 0, device=dev)
gx, gw, ga = pcf_cuda.pconv_backward(d(g['gcap.agg']), d(a['dense_feats']), d(a['nei_inds']), d(g['cap.w']), add)
print('direct gx', (gx.cpu() - g['gin.dense_feats']).abs().max().item())
gx2, _, _ = pcf_cuda.pconv_backward(d(g['gcap.agg']), d(a['dense_feats']), d(a['nei_inds']), caps['w'].detach().contiguous(), add)
print('direct gx with fused w', (gx2.cpu() - g['gin.dense_feats']).abs().max().item())
for name, p in layer.named_parameters():
    print(name, (p.grad.cpu() - g['gsd.' + name]).abs().max().item())
