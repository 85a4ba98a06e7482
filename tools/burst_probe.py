import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_vllm_amd.config import Config
from tools.synthetic import SyntheticDecodeDriver as SparseDecodeDriver
from sparse_vllm_amd.utils.profiler import profiler
B=64
cfg = Config.from_kwargs(sparse_method="h2o", h2o_decode_budget=4096, h2o_decode_eviction_interval=128, h2o_prefill_budget=8192,
                         max_model_len=4224 + 64, max_num_seqs_in_gpu=B, num_kvcache_slots=B * 4224 + 4096)
drv = SparseDecodeDriver(cfg)
drv.cache_manager.permute_free_slots(1)
drv.admit_resident_rows(B, 4096, logical_len=131072, seed=0, device_rng=True)
q, k, v = drv.random_step_inputs(seed=1)
drv.enable_decode_graph()
ts=[]
for i in range(300):
    torch.cuda.synchronize(); t0=time.perf_counter()
    drv.step(q,k,v)
    torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
ts=np.array(ts)
big=np.argsort(-ts)[:6]
print("median %.3f ms; top:"%np.median(ts), [(int(i), round(float(ts[i]),2)) for i in sorted(big)])
print("mean over 128..255: %.3f" % ts[128:256].mean())
