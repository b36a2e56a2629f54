"""A/B of the Lloyd step: run once with NIF_KMEANS_LLOYD_H=1 (the step that materialises H) and once without (the
fused pass), `python tools/lloyd_ab.py dump FILE` each, then `python tools/lloyd_ab.py cmp FILE_A FILE_B`."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def problems():
    for N, k, m, seed in [(50, 7, 40, 1), (300, 64, 64, 2), (1000, 130, 128, 3), (2500, 16, 96, 4), (6144, 768, 256, 5),
                          (24576, 3072, 64, 6)]:
        g = torch.Generator().manual_seed(seed)
        A = torch.randn(m, N, generator=g)
        s = torch.rand(N, generator=g) + 0.2
        assign = torch.randint(0, k, (N,), generator=g).int()
        assign[assign == 1] = 2
        yield N, k, A, s, assign


def dump(path):
    from neuron_interpolation_model_fusion_amd import ops
    out = []
    for N, k, A, s, assign in problems():
        G, _, _ = ops.xty(A.double().cuda(), None, acc='f64', norms=False)
        km = ops.KMeans(G, torch.diagonal(G).contiguous(), s.cuda(), k)
        a = assign.cuda()
        for _ in range(3):
            a, nearest, cost, kp = km.lloyd_step(a)
            out.append((a.cpu(), nearest.cpu(), cost.cpu(), kp.cpu()))
    torch.save(out, path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    assert len(A) == len(B)
    for x, y in zip(A, B):
        for u, v in zip(x, y):
            assert torch.equal(u, v), (u, v)
    print(f'{len(A)} Lloyd steps: labels, distances, costs and k\' bit-identical')


if __name__ == '__main__':
    if sys.argv[1] == 'dump':
        dump(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
