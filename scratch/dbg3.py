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
layer = pcf_layers.PointConv(3, 32, c, weightnet=[3, 16])
layer.load_state_dict(split(g, 'sd.'), strict=True); layer.to(dev).train()
feats = a['dense_feats'].to(dev).requires_grad_(True)
out, wn = layer(a['dense_xyz'].to(dev), feats, a['nei_inds'].to(dev))
gup = g['gup'].to(dev)
P = dict(layer.named_parameters())
def chk(tag, inputs, names):
    gr = torch.autograd.grad(out, inputs, gup, retain_graph=True)
    torch.cuda.synchronize()
    for n, t in zip(names, gr):
        ref = g['gin.dense_feats'] if n == 'feats' else g['gsd.' + n]
        print(tag, n, (t.cpu() - ref).abs().max().item())
chk('only-top', [feats, P['linear.bias']], ['feats', 'linear.bias'])
chk('+wn2', [feats, P['linear.bias'], P['weightnet.mlp_convs.2.c.weight']], ['feats', 'linear.bias', 'weightnet.mlp_convs.2.c.weight'])
chk('+wn1', [feats, P['linear.bias'], P['weightnet.mlp_convs.1.c.weight']], ['feats', 'linear.bias', 'weightnet.mlp_convs.1.c.weight'])
chk('+wn0', [feats, P['linear.bias'], P['weightnet.mlp_convs.0.c.weight']], ['feats', 'linear.bias', 'weightnet.mlp_convs.0.c.weight'])
