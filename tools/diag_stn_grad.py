import sys, json, numpy as np, torch
sys.path.insert(0, '.')
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
from fudanocr_amd import kernels as K
arch = sys.argv[1] if len(sys.argv) > 1 else 'tsrn'
g = np.load('tests/golden/%s_train_mse.npz' % arch)
gn = json.load(open('tests/golden/%s_train_mse_gradnorms.json' % arch))
net, _, _ = build_models(torch.device('cuda:0'), arch)
net.train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout): m.eval()
lr, hr, _ = make_batch(4, 1234)
# capture ctrl points and their grad
ctrl_box = {}
orig = net.stn_head.forward
def hooked(x):
    f, c = orig(x); c.retain_grad(); ctrl_box['c'] = c; return f, c
net.stn_head.forward = hooked
sr = net(lr.cuda()); mse = K.mse_loss(sr, hr.cuda()); (mse*100).backward()
P = dict(net.named_parameters())
print('sr rel', float((sr.cpu()-torch.tensor(g['sr'])).abs().max()/np.abs(g['sr']).max()))
for k,v in sorted(gn.items()):
    if v is None: continue
    gv = float(P[k].grad.norm())
    if abs(gv-v) > 1e-2*v + 1e-5: print('NORM', k, gv, v)
# oracle with ctrl grads
from oracle import sr_oracle as O
from fudanocr_amd.utils.weight_fill import fill_dict_
Pm = O.make_params(O.schema_sr(arch)); fill_dict_({k:v.data for k,v in Pm.items()})
x = lr.clone()
c0 = O.stn_head(Pm, x, True); c0.retain_grad()
xw = O.tps_warp(Pm, x, c0)
# run the rest by monkeypatching: call sr_forward with stn False on warped input in training mode
srm = O.sr_forward(Pm, arch, xw, True, stn=False)
(((srm-hr)**2).mean()*100).backward()
print('ctrl diff', (ctrl_box['c'].detach().cpu()-c0.detach()).abs().max().item())
dg = ctrl_box['c'].grad.cpu(); do = c0.grad
print('dctrl rel-to-max', ((dg-do).abs().max()/do.abs().max()).item())
print('per-sample dctrl err', [(dg[b]-do[b]).abs().max().item() for b in range(4)], 'max', do.abs().max().item())
