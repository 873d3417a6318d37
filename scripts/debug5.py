import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import mpc_gpu
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
seq, noise = GOLD["noisy_seq"], GOLD["noisy_noise"]
dev = torch.device("cuda:0")
with mpc_gpu.BatchedMpc(20, 5, 2.0, max_batch=2) as s:
    for k in range(30):
        st = torch.from_numpy(seq[:, k].copy()).to(dev); nz = torch.from_numpy(noise[:, k].copy()).to(dev)
        torch.cuda.synchronize()
        s.obstacle_step_dev(8, st, nz, 0.1, 2.0)
        torch.cuda.synchronize()
        d = st.cpu().numpy() - seq[:, k + 1]
        if np.abs(d).max() > 0:
            i, j = np.unravel_index(np.abs(d).argmax(), d.shape)
            print("step", k, "max diff", np.abs(d).max(), "at", i, j, "in", seq[i, k], "noise", noise[i, k], "want", seq[i, k+1], "got", st.cpu().numpy()[i])
            break
    else:
        print("all equal")
