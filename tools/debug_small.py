import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "level-s2fm_official_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ls2fm.options import make_options
from ls2fm.models.SDF import SDF
from ls2fm.models.RadF import RadF
from ls2fm.models.Renderer import Renderer
from ls2fm import fused
for n_samples in (1, 2, 3, 5):
    opt = make_options("BlendedMVS", device="cuda", dual_field=True, sample_intvs=n_samples,
                       hash_encoding=dict(n_levels=6, n_features_per_level=2, log2_hashmap_size=12, base_resolution=16))
    torch.manual_seed(0)
    sdf, rad, ren = SDF(opt).cuda(), RadF(opt).cuda(), Renderer(opt)
    c = torch.tensor([0., 0., -5.]).repeat(1, 20, 1).cuda(); d = (torch.tensor([0., 0., 1.]).repeat(1, 20, 1) + 0.1 * torch.randn(1, 20, 3)).cuda()
    print("N", n_samples, "can_render", fused.can_render(ren, opt, c, d, sdf, rad))
    try:
        ret = ren.forward(opt, c, d, sdf, rad)
        torch.cuda.synchronize()
        print(" rgb[0]", ret["rgb"][0, 0].tolist(), "depth", ret["depth_mlp"][0, 0].tolist())
        ret["rgb"].sum().backward(); torch.cuda.synchronize()
        gf = {**{"s." + k: p.grad for k, p in sdf.named_parameters()}, **{"r." + k: p.grad for k, p in rad.named_parameters()}}
        sdf.zero_grad(); rad.zero_grad()
        ret = ren.forward_composed(opt, c, d, sdf, rad)
        ret["rgb"].sum().backward(); torch.cuda.synchronize()
        for k, p in list(sdf.named_parameters()) + list(rad.named_parameters()):
            pass
        gc = {**{"s." + k: p.grad for k, p in sdf.named_parameters()}, **{"r." + k: p.grad for k, p in rad.named_parameters()}}
        for k in gf:
            a, b = gf[k], gc[k]
            if a is None or b is None:
                print("  ", k, "None", a is None, b is None); continue
            nan = int((~torch.isfinite(a)).sum())
            err = float((a - b).abs().max() / (b.abs().max() + 1e-30)) if nan == 0 else float("nan")
            print("  %-40s nan %6d  rel %.2e  |ref| %.2e" % (k, nan, err, float(b.abs().max())))
    except Exception as e:
        print(" EXC", repr(e)[:300])
