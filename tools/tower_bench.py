"""Images/s of the CLIP ViT-B/16 image tower and of the DINO / GCD ViT-B/16 tower on the same box, same launch size (round 6: the
DINO tower is --feat_model dino_vit of both shipped scripts, main_unsup.py:240-255; its fc1 epilogue is exact GELU).

  python tools/tower_bench.py [launches] [batch]      -> one JSON line; fc1 launches bracketed by HIP events (scd_encoder_timing)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scd_amd.clip import weights as W                      # noqa: E402
from scd_amd.clip.model import CLIP, DinoViT               # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 3990
    which = sys.argv[3] if len(sys.argv) > 3 else "both"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((batch, 3, 224, 224), device=dev, generator=g, dtype=torch.float32).half()
    towers = []
    if which in ("both", "clip"):
        clip = CLIP(W.synthetic_clip_state_dict(seed=0, text=False)).cuda()
        towers.append(("clip", clip.visual.enc))
    if which in ("both", "dino"):
        dino = DinoViT(W.synthetic_dino_state_dict(seed=1, layers=12)).cuda()
        towers.append(("dino", dino._enc))
    out = {"batch": batch, "launches": launches, "lib": os.environ.get("SCD_HIP_LIB", "default")}
    for rep in range(2):
        for name, enc in towers:
            enc.encode_image(x)
            torch.cuda.synchronize()
            enc.timing(True)
            t0 = time.perf_counter()
            for _ in range(launches):
                enc.encode_image(x)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ms, n, flop = enc.timing(False)
            out["%s_rep%d" % (name, rep)] = {"images_per_s": round(batch * launches / dt, 1), "fc1_us_per_launch": round(ms * 1e3 / max(n, 1), 1),
                                            "fc1_frac_of_2.5PF": round(flop / max(ms, 1e-9) / 1e9 / 2500.0, 4), "fc1_launches": n}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
