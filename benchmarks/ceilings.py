#!/usr/bin/env python3
"""Empirical ceilings of the box: a stream copy / fill / read and a large hipBLASLt bf16 GEMM, next to the
vendor figures (8 TB/s HBM3E, 2.5 PFLOP/s dense bf16) the roofline fractions are quoted against."""
import torch


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e-3


def main():
    n = 1 << 29                                       # 2 GiB of fp32
    a, b = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    t = timeit(lambda: b.copy_(a))
    print(f"copy  2 GiB -> 2 GiB : {2 * 4 * n / t / 1e12:5.2f} TB/s (read + write)")
    t = timeit(lambda: a.zero_())
    print(f"fill  2 GiB          : {4 * n / t / 1e12:5.2f} TB/s (write)")
    t = timeit(lambda: a.sum())
    print(f"sum   2 GiB          : {4 * n / t / 1e12:5.2f} TB/s (read)")
    for m in (4096, 8192, 16384):
        x = torch.randn(m, m, device="cuda", dtype=torch.bfloat16)
        y = torch.randn(m, m, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: torch.mm(x, y), reps=5)
        print(f"bf16 GEMM {m}^3      : {2 * m ** 3 / t / 1e12:6.0f} TFLOP/s")


if __name__ == "__main__":
    main()
