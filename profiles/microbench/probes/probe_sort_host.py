"""host cost of one rp_sort_pairs_i32 call (enqueue only, device kept busy but not waited on), by number of digit passes"""
import time, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rec_pangu_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1703936
keys = torch.randint(0, 33762603, (n,), dtype=torch.int32, device="cuda")
nbytes = C.c_size_t(0)
hip.lib().rp_sort_workspace_bytes(n, C.byref(nbytes))
ws = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
ko, po = torch.empty_like(keys), torch.empty_like(keys)
st = torch.cuda.current_stream().cuda_stream
for eb in (9, 18, 26, 32):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            hip.lib().rp_sort_pairs_i32(ws.data_ptr(), nbytes.value, keys.data_ptr(), ko.data_ptr(), po.data_ptr(), n, eb, st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"RP_SORT={os.environ.get('RP_SORT','own')} n={n} end_bit={eb}: enqueue {(t1-t0)/200*1e6:.1f} us/call, with device {(t2-t0)/200*1e6:.1f} us/call", flush=True)
# the python wrapper (workspace query + torch.empty x3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    hip.sort_pairs(keys, end_bit=26)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"hip.sort_pairs wrapper: enqueue {(t1-t0)/200*1e6:.1f} us/call")
