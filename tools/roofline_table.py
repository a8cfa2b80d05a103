"""Algorithmic FLOPs / bytes per image of every kernel family of the training step (bf16 storage),
and the roof that bounds it on MI355X (ridge = 2.5 PFLOP/s / 8 TB/s = 312 flop/B)."""
S = 192
convs = [  # name, Hin, Cin, Cout, k, s, count
    ("conv1 3->64 s2", 192, 3, 64, 3, 2, 1), ("conv2 64->128 s2", 96, 64, 128, 3, 2, 1),
    ("s1.cv1 1x1 128->128", 48, 128, 128, 1, 1, 1), ("s1 3x3 64->64", 48, 64, 64, 3, 1, 4), ("s1.cv4 1x1 256->128", 48, 256, 128, 1, 1, 1),
    ("down1 128->256 s2", 48, 128, 256, 3, 2, 1),
    ("s2.cv1 1x1 256->256", 24, 256, 256, 1, 1, 1), ("s2 3x3 128->128", 24, 128, 128, 3, 1, 4), ("s2.cv4 1x1 512->256", 24, 512, 256, 1, 1, 1),
    ("down2 256->512 s2", 24, 256, 512, 3, 2, 1),
    ("s3.cv1 1x1 512->512", 12, 512, 512, 1, 1, 1), ("s3 3x3 256->256", 12, 256, 256, 3, 1, 4), ("s3.cv4 1x1 1024->512", 12, 1024, 512, 1, 1, 1),
    ("proj 1x1 512->256", 12, 512, 256, 1, 1, 1),
]
print(f"{'layer':24s} {'MFLOP/img':>10s} {'KB/img':>8s} {'flop/B':>7s} bound")
tf = tb = 0
for name, h, ci, co, k, s, n in convs:
    ho = h // s
    fl = 2 * ho * ho * co * k * k * ci
    esz = 4 if ci == 3 else 2
    by = h * h * ci * esz + ho * ho * co * 2
    print(f"{name:24s} {fl*n/1e6:10.1f} {by*n/1e3:8.1f} {fl/by:7.0f} {'mfma' if fl/by > 312 else 'hbm'}")
    tf += fl * n
    tb += by * n
N, D = 145, 256
lin = [("qkv 256->768", 256, 768, 4), ("to_out 256->256", 256, 256, 4), ("fc1 256->256", 256, 256, 4), ("fc2 256->256", 256, 256, 4)]
for name, k, n_, cnt in lin:
    fl = 2 * N * k * n_
    by = N * (k + n_) * 2
    print(f"{name:24s} {fl*cnt/1e6:10.1f} {by*cnt/1e3:8.1f} {fl/by:7.0f} {'mfma' if fl/by > 312 else 'hbm'}")
    tf += fl * cnt
    tb += by * cnt
fl = 2 * 2 * 8 * N * N * 32
by = N * 768 * 2 + N * 256 * 2
print(f"{'attention (per layer)':24s} {fl/1e6:10.1f} {by/1e3:8.1f} {fl/by:7.0f} hbm   x4")
tf += 4 * fl
tb += 4 * by
fl = 2 * 48 * 48 * 256 * 21
by = 144 * 256 * 2 + 21 * 48 * 48 * 4
print(f"{'pose head':24s} {fl/1e6:10.1f} {by/1e3:8.1f} {fl/by:7.0f} hbm")
print(f"forward total {tf/1e9:.3f} GFLOP/img, {tb/1e6:.2f} MB/img (GEMM-shaped ops only)")
