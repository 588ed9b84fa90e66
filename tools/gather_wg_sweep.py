import os, sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
data = bench.synthetic_rows(50000, 5000, 546, dev)
for wg in (64, 96, 128, 192, 256, 512, 0):
    os.environ["MMVAE_GATHER_WG"] = str(wg)
    r = bench.data_path(data, 2, 5000, 5000)
    print(wg, round(r["shuffled_epoch_ms_per_step_pipelined"], 4), round(r["shuffled_epoch_ms_per_step_back_to_back"], 4), flush=True)
