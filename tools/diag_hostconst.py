import sys, numpy as np, torch
sys.path.insert(0, ".")
from fudanocr_amd.model.tps_spatial_transformer import TPSSpatialTransformer
from fudanocr_amd.model.tbsrn import positionalencoding2d
u = np.load("tests/golden/units.npz")
m = TPSSpatialTransformer(output_image_size=(16, 64), num_control_points=20, margins=(0.05, 0.05))
d = np.abs(m.inverse_kernel.numpy() - u["tps_inv"])
print("inverse_kernel: max abs diff %.3e (max |ref| %.1f), entries differing %d / %d" % (d.max(), np.abs(u["tps_inv"]).max(), (d > 0).sum(), d.size))
d = np.abs(m.target_coordinate_repr.numpy() - u["tps_repr"])
print("coordinate_repr: max abs diff %.3e, entries differing %d / %d" % (d.max(), (d > 0).sum(), d.size))
d = np.abs(positionalencoding2d(64, 16, 64).numpy() - u["pe"])
print("positional enc : max abs diff %.3e, entries differing %d / %d" % (d.max(), (d > 0).sum(), d.size))
