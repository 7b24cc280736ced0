"""kernel timeline of one captured stage step (ls2fm.stage.RenderStage, benchmark shape):
     rocprofv3 --kernel-trace -d gpurun_out/tl -- python tools/stage_timeline.py run ; python tools/stage_timeline.py show gpurun_out/tl"""
import glob, os, sqlite3, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "level-s2fm_official_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run():
    import torch
    import bench
    from ls2fm import stage
    from ls2fm.options import make_options
    from ls2fm.models.SDF import SDF
    from ls2fm.models.RadF import RadF
    from ls2fm.models.Renderer import Renderer
    dev = "cuda"
    opt = make_options("ETH3D", device=dev, dual_field=True, sample_intvs=128)
    torch.manual_seed(0)
    sdf, rad, ren = SDF(opt).to(dev), RadF(opt).to(dev), Renderer(opt)
    bench.randomize([sdf, rad])
    center, ray = bench.synthetic_rays(1024, 5.0, dev)
    gt = torch.rand(1, 1024, 3, device=dev)
    st = stage.RenderStage(opt, ren, sdf, rad, weights=dict(rgb=3, eikonal_loss=2, DC_Loss=0), lr=1e-3, lr_end=1e-4, max_iter=1000,
                           capture=True)
    with torch.cuda.stream(torch.cuda.Stream()):
        for _ in range(30):
            st.step(center, ray, gt)
        torch.cuda.synchronize()


def show(path):
    db = glob.glob(path + '/*/*.db')[0]
    cur = sqlite3.connect(db).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tables if 'kernel_dispatch' in t and not t.startswith('rocpd_info')][0]
    ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    marks = [i for i, r in enumerate(rows) if 'sphere_trace' in r[3]]
    a, b = marks[-3], marks[-2]
    t0 = rows[a][0]
    for st, en, q, name in rows[a:b]:
        name = name.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN2at6native", "at::")
        print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{q}  {name[:70]}")
    print("step span us (under the profiler):", (rows[b][0] - t0) / 1e3)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
