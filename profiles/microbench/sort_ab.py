"""pair sort timing at the Criteo shape (1.7 M (arena row, position) pairs, 26 key bits); RP_SORT_BITS selects the digit width"""
import sys, torch
sys.path.insert(0, '.')
import bench
from rec_pangu_amd import hip
hip.lib()
enc = bench.criteo_enc_dict(1)
b = bench.synth_batch(enc, 65536, 5, 'cuda')
rows = [v['vocab_size'] + 1 for v in enc.values() if 'vocab_size' in v]
base = torch.tensor([0] + list(torch.tensor(rows).cumsum(0)[:-1]), device='cuda')
keys = (torch.stack([b[f'C{i+1}'] for i in range(26)]) + base[:, None]).reshape(-1).to(torch.int32)
ref_k, ref_p = torch.sort(keys.long(), stable=True)
ko, po = hip.sort_pairs(keys, end_bit=26)
assert torch.equal(ko.long(), ref_k) and torch.equal(po.long(), ref_p), "sort wrong"
for _ in range(3): hip.sort_pairs(keys, end_bit=26)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): hip.sort_pairs(keys, end_bit=26)
e1.record(); torch.cuda.synchronize()
import os
print("RP_SORT_BITS", os.environ.get("RP_SORT_BITS"), f"{e0.elapsed_time(e1)/20*1000:.1f} us")
