import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd.aster import AsterInferer
torch.manual_seed(0)
dev = torch.device('cuda:0')
o_cpu = AsterInferer(); o_gpu = AsterInferer().to(dev); o64 = AsterInferer().double()
x = torch.randn(4, 3, 64, 256) * 0.5
labels = torch.tensor([[5,6,1,1,1,1,1,1],[2,3,4,5,6,7,8,9],[7,1,1,1,1,1,1,1],[3,4,5,6,1,1,1,1]])
def run(o, x, labels):
    x = x.clone().requires_grad_(True)
    inp = o.convert_inputs(x, labels)
    m = o.model
    img = inp.permute(0,3,1,2)
    r = m.rectify(img); f = m.resnet(m.stem(r)); seq = f.squeeze(2).permute(0,2,1); enc,_ = m.rnn(seq)
    logits = o(inp)
    ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1).to(x.device), reduction='sum')
    (g,) = torch.autograd.grad(ce, x)
    return dict(rect=r, feat=f, enc=enc, logits=logits, ce=ce, grad=g, argmax=logits.argmax(2))
ref = run(o64, x.double(), labels)
a = run(o_cpu, x, labels)
b = run(o_gpu, x.to(dev), labels.to(dev))
def rel(u, v): 
    u=u.detach().double().cpu(); v=v.detach().double().cpu(); return float((u-v).abs().max()/(v.abs().max()+1e-30))
for k in ('rect','feat','enc','logits','ce','grad'):
    print(k, 'cpu32 vs 64: %.3e' % rel(a[k], ref[k]), ' gpu32 vs 64: %.3e' % rel(b[k], ref[k]), ' gpu vs cpu32: %.3e' % rel(b[k], a[k]))
print('argmax equal gpu/cpu/64:', bool((a['argmax']==b['argmax'].cpu()).all()), bool((a['argmax']==ref['argmax']).all()))
lg = ref['logits']; top2 = lg.topk(2, dim=2).values; print('min top1-top2 gap', float((top2[...,0]-top2[...,1]).min()), 'logit scale', float(lg.abs().max()))
print('allow_tf32 conv', torch.backends.cudnn.allow_tf32, 'matmul', torch.backends.cuda.matmul.allow_tf32)
