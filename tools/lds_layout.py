#!/usr/bin/env python3
"""LDS bank model of the exchanges of spectrum_kernel / spectrum_pair_kernel (no GPU needed).

Replays every ds_write / ds_read wave-instruction of one window against the banking rules of
/opt/skills/guides/MI355X_MICROARCH.md (LDS): ds_write_b64 is served in 4 groups of 16 consecutive lanes over 32 banks
(~6 cycles per instruction for the VGPR -> LDS transfer: a conflict only costs once the array cycles exceed that),
ds_read_b64 in 2 groups of 32 lanes over 64 banks, ds_write_b128 in 8 groups of 8 lanes over 32 banks (13 cycles of
transfer), ds_read_b128 in 4 groups of 16 lanes (the guide's lane sets) over 64 banks; every extra distinct address on a
busy bank adds one cycle to its group.  Prints, per transform size, the LDS cycles per wave and window of the
one-pad-per-16 natural order (rounds 1-3) and of the transposed / natural layout of Plan<N> (round 4), with the
conflict-free ideal beside them.  The address formulas below are the kernels' own (ksa_kernels.hpp, ksa_kernels_pair.hpp).
"""
def ilog2(n):
    return n.bit_length() - 1


def plan(N):
    log2n = ilog2(N)
    M = (log2n + 3) // 4
    R0 = 1 << (log2n - 4 * (M - 1))
    L = N // 16
    T = max(64, L)
    P = dict(N=N, M=M, R0=R0, B0=16 // R0, L=L, T=T, S=T // L, NB=N // R0)
    P["ST1"] = 17 if P["NB"] == 16 else P["NB"] + 32 // R0
    P["SH2"], P["K2"] = 4 + ilog2(R0), R0 & 15
    x1 = (R0 - 1) * P["ST1"] + P["NB"]
    x2 = N + P["K2"] * ((N - 1) >> P["SH2"]) if M == 3 else 0
    P["NPAD"] = (max(x1, x2) + 1) & ~1
    return P


def perm(R, p):
    return ((p >> 2) | ((p & 3) << 2)) if R == 16 else (((p >> 1) | ((p & 1) << 2)) if R == 8 else p)


B128_READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_READ_GROUPS += [[x + 32 for x in g] for g in B128_READ_GROUPS]
GROUPS = {"w64": ([list(range(16 * k, 16 * k + 16)) for k in range(4)], 16, 6), "r64": ([list(range(32 * k, 32 * k + 32)) for k in range(2)], 32, 0),
          "w128": ([list(range(8 * k, 8 * k + 8)) for k in range(8)], 8, 13), "r128": (B128_READ_GROUPS, 16, 0)}


def cycles(addrs, kind):
    """addrs: element index per lane (elements are the access width); returns the instruction's cycles."""
    groups, mod, floor = GROUPS[kind]
    tot = 0
    for g in groups:
        banks = {}
        for ln in g:
            banks.setdefault(addrs[ln] % mod, set()).add(addrs[ln])
        tot += max(len(v) for v in banks.values())
    return max(tot, floor)


def window_cycles(N, layout, pair=False):
    """LDS cycles of the data exchanges per wave and window.  layout: 'pad16' | 'x'."""
    P = plan(N)
    L, T, R0, B0, M = P["L"], P["T"], P["R0"], P["B0"], P["M"]
    ST1, K2, SH2, NPAD = (260, 4, 6, 1084) if pair else (P["ST1"], P["K2"], P["SH2"], P["NPAD"])
    if layout == "pad16":
        NPAD = N + N // 16
    padi = lambda i: i + (i >> 4)
    w, r = ("w128", "r128") if pair else ("w64", "r64")
    tot = 0
    for wave in range(T // 64):
        lanes = range(64 * wave, 64 * wave + 64)
        sl = [(tid // L, tid % L) for tid in lanes]
        for b in range(B0):
            for t in range(R0):
                if layout == "x":
                    a = [s * NPAD + perm(R0, t) * ST1 + (l + b * L) for s, l in sl]
                else:
                    a = [s * NPAD + padi((l + b * L) * R0 + perm(R0, t)) for s, l in sl]
                tot += cycles(a, w)
        pp = R0
        for s_ in range(1, M):
            for t in range(16):
                if layout != "x":
                    a = [s * NPAD + padi(l + L * t) for s, l in sl]
                elif s_ == 1:
                    a = [s * NPAD + (l % R0) * ST1 + l // R0 + (L // R0) * t for s, l in sl]
                else:
                    a = [s * NPAD + l + (L + K2) * t for s, l in sl]
                tot += cycles(a, r)
            if s_ < M - 1:
                for t in range(16):
                    a = []
                    for s, l in sl:
                        kk = l & (pp - 1)
                        j = (l - kk) * 16 + kk
                        a.append(s * NPAD + (j + K2 * (l >> ilog2(R0)) + perm(16, t) * R0 if layout == "x" else padi(j + perm(16, t) * pp)))
                    tot += cycles(a, w)
                pp *= 16
    return tot / (T // 64)


if __name__ == "__main__":
    print("%-22s %10s %10s %10s" % ("kernel", "pad16", "round 4", "ideal"))
    for N in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        P = plan(N)
        ideal = (16 * 6 + 16 * 2) * (P["M"] - 1)
        print("spectrum_kernel<%d>%s %10.0f %10.0f %10d   (NPAD %d, ST1 %d, K2 %d per %d)" % (
            N, " " * (5 - len(str(N))), window_cycles(N, "pad16"), window_cycles(N, "x"), ideal, P["NPAD"], P["ST1"], P["K2"], 1 << P["SH2"]))
    print("spectrum_pair_kernel<1024> %6.0f %10.0f %10d   (16-byte elements: NPAD 1084, ST1 260, K2 4 per 64)" % (
        window_cycles(1024, "pad16", True), window_cycles(1024, "x", True), (16 * 13 + 16 * 4) * 2))
