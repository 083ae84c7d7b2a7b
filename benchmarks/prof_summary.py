#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv by kernel class: launches and ms per train step."""
import collections
import csv
import sys


def klass(n):
    if n.startswith('Cijk') or n.startswith('Custom_Cijk') or 'gemm' in n.lower(): return 'gemm (hipBLASLt)'
    if '_ZN2bg' in n or 'bg::' in n:
        for k in ('adam_step', 'embed_', 'attn_bwd_one', 'attn_bwd_dq', 'attn_bwd_dkv', 'attn_bwd_slots', 'attn_fwd', 'attn_weights', 'layernorm_bwd', 'add_layernorm_fwd', 'bias_act_bwd',
                  'partial_reduce', 'softmax_stats', 'topn', 'topp', 'loss', 'rope'):
            if k in n: return 'bg:' + k
        return 'bg:other'
    if 'layer_norm' in n or 'GammaBeta' in n: return 'torch layernorm'
    if 'multi_tensor' in n or 'FusedAdam' in n or 'fused_adam' in n.lower(): return 'torch adam/foreach'
    if 'reduce_kernel' in n: return 'torch reduce'
    if 'CatArrayBatchedCopy' in n: return 'torch cat (dW operand stacking)'
    if 'copy' in n.lower(): return 'torch copy/cast'
    if 'Fill' in n or 'fillBuffer' in n: return 'torch fill'
    if 'CUDAFunctor_add' in n or 'AddFunctor' in n: return 'torch add'
    if 'Gelu' in n: return 'torch gelu'
    if 'embedding' in n: return 'torch embedding'
    if 'elementwise' in n: return 'torch elementwise other'
    return 'other'


rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cat = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = klass(r['Name'])
    cat[k][0] += int(r['Calls'])
    cat[k][1] += float(r['TotalDurationNs'])
tot = sum(v[1] for v in cat.values())
print(f"{'class':28s} {'launches/step':>14s} {'ms/step':>9s} {'avg us':>8s} {'share':>6s}")
for k, (n, t) in sorted(cat.items(), key=lambda x: -x[1][1]):
    print(f"{k:28s} {n / steps:14.1f} {t / 1e6 / steps:9.3f} {t / 1e3 / n:8.1f} {100 * t / tot:5.1f}%")
print(f"{'total':28s} {sum(v[0] for v in cat.values()) / steps:14.1f} {tot / 1e6 / steps:9.3f}")
