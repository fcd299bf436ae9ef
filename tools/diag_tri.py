import sys, numpy as np, torch
sys.path.insert(0, ".")
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
from fudanocr_amd.utils.weight_fill import fill_dict_
from oracle import sr_oracle as O
lr, hr, _ = make_batch(4, 1234)
g = torch.tensor(np.load("tests/golden/tbsrn_train_mse.npz")["sr"]).double()
pe0 = O.positional_encoding_2d
O.positional_encoding_2d = lambda *a: pe0(*a).double()
P = O.make_params(O.schema_sr("tbsrn")); fill_dict_({k: v.data for k, v in P.items()})
P = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in P.items()}
with torch.no_grad(): t = O.sr_forward(P, "tbsrn", lr.double(), True)
O.positional_encoding_2d = pe0
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
print("golden vs truth %.3e" % rel(g, t))
for with_crnn in (False, True):
    for nograd in (True, False):
        net, _, _ = build_models(torch.device("cuda:0"), "tbsrn", with_crnn=with_crnn)
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout): m.eval()
        if nograd:
            with torch.no_grad(): o = net(lr.cuda()).cpu().double()
        else:
            o = net(lr.cuda()).detach().cpu().double()
        print("with_crnn=%s no_grad=%s: gpu vs truth %.3e   gpu vs golden %.3e" % (with_crnn, nograd, rel(o, t), rel(o, g)))
