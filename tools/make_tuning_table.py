"""Dev tool (GPU box): measure the launches of the BASELINE workloads and write the table the library ships with.

    python tools/make_tuning_table.py gpurun_out/gfx950.txt      # then copy to ccvpe_amd/tuning/gfx950.txt and commit

Every workload of bench.py at batch 32 and batch 1 (plus the test suite's golden configurations at their batch sizes), fp32 and
bf16x3, full forward and the cached-aerial pair.  Starts from an empty table (the committed one and the user cache are ignored), so
every entry is a fresh measurement of this build on this device."""
import os
import sys

os.environ["CCVPE_TUNE_IGNORE_COMMITTED"] = "1"
os.environ["CCVPE_TUNE_CACHE"] = "off"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from ccvpe_amd import models, tuning, weights

WORK = [   # variant, ctor kwargs, fov, batches
    ("vigor_ori_prior", dict(ori_noise=180.0, circular_padding=True), 360.0, (32, 1, 2)),
    ("vigor_ori_prior", dict(ori_noise=72.0, circular_padding=False), 108.0, (32, 1)),
    ("vigor", dict(circular_padding=True), 360.0, (1,)),
    ("kitti", {}, 360.0, (32, 1)),
    ("oxford", {}, 360.0, (1, 2)),
]


def build(variant, kw, precision):
    cls = {"vigor": models.CVM_VIGOR, "vigor_ori_prior": models.CVM_VIGOR_ori_prior, "kitti": models.CVM_KITTI, "oxford": models.CVM_OxfordRobotCar}[variant]
    if variant == "vigor":
        m = cls("cuda", kw["circular_padding"], precision=precision)
    elif variant == "vigor_ori_prior":
        m = cls("cuda", kw["ori_noise"], kw["circular_padding"], precision=precision)
    else:
        m = cls("cuda", precision=precision)
    m.load_state_dict(weights.generate_state_dict(variant, 0))
    return m.to("cuda").eval()


def device_string():
    """Marketing name when the driver knows one ("AMD Radeon Graphics" on unnamed engineering boards), always with the ISA and CU count."""
    pr = torch.cuda.get_device_properties(0)
    arch = getattr(pr, "gcnArchName", "").split(":")[0]
    return f"{pr.name} ({arch}, {pr.multi_processor_count} CUs, {pr.total_memory / 2**30:.0f} GiB)"


def main():
    out = sys.argv[1]
    table = {}
    for variant, kw, fov, batches in WORK:
        for precision in ("fp32", "bf16x3"):
            if precision == "bf16x3" and 32 not in batches:
                continue
            m = build(variant, kw, precision)
            for b in batches:
                if precision == "bf16x3" and b != 32:
                    continue
                g, s = weights.generate_inputs(variant, b, 0, fov)
                g, s = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda()
                m(g, s)
                if precision == "fp32" and b <= 2:
                    m.forward_cached(g, m.encode_aerial(s))
                torch.cuda.synchronize()
                print(variant, kw, precision, b, "ok", flush=True)
            table.update(tuning.parse(m.export_tuning()))
            del m
            torch.cuda.empty_cache()
    with open(out, "w") as fh:
        fh.write("# tuning table of ccvpe_amd (see ccvpe_amd/tuning.py); measured by tools/make_tuning_table.py on " + device_string() + "\n")
        fh.write(tuning.render(table))
    print(len(table), "launches ->", out)


if __name__ == "__main__":
    main()
