#!/usr/bin/env python3
"""Where do the path kernel's LDS bank conflicts come from?  (SQ_LDS_BANK_CONFLICT is 18 % of SQ_LDS_IDX_ACTIVE and no ledger
mentions banks.)  No counter attributes conflicts to instructions, so this is a MODEL: the banking rules of /opt/skills/guides/MI355X_MICROARCH.md
(section LDS: lane groups and bank function per instruction; every extra distinct address on a busy bank within a group costs one LDS cycle)
applied by Monte Carlo to the kernel's LDS reads whose address differs from lane to lane, with the blob's real offsets and strides (csrc/ssx_blob.h)
and the lane populations of profiles/r05/lanestat.log.  It is checked against the one thing that can be measured: the change of the counter when
the quad records' stride changes (40 -> 44 words, build -DSSX_QUAD_PAD_WORDS=4: profiles/r06/ab_third_call.log).
    python tools/lds_conflict_model.py [--quads 19 --trials 4000]"""
import argparse
import random

# lane groups per instruction (guide): list of lane lists; bank modulus
G_B32 = ([list(range(0, 32)), list(range(32, 64))], 32)
G_B64 = ([list(range(0, 32)), list(range(32, 64))], 64)
def _g128():
    base = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    return base + [[l + 32 for l in g] for g in base]
G_B128 = (_g128(), 64)
def _g96():
    base = [[0, 1, 2, 3, 20, 21, 22, 23], [4, 5, 6, 7, 16, 17, 18, 19], [8, 9, 10, 11, 28, 29, 30, 31], [12, 13, 14, 15, 24, 25, 26, 27]]
    return base + [[l + 32 for l in g] for g in base]
G_B96 = (_g96(), 32)


def cycles(groups, addrs, width_dwords):
    """(base cycles, extra cycles): addrs[lane] = byte address or None (lane inactive).  Per group one cycle + one per extra distinct address on a busy bank."""
    lane_groups, mod = groups
    base = extra = 0
    for g in lane_groups:
        banks = {}
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            for d in range(width_dwords):
                banks.setdefault(((a // 4) + d) % mod, set()).add((a // 4) + d)
        if banks:
            base += 1
            extra += max(len(v) for v in banks.values()) - 1
    return base, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quads", type=int, default=19)
    ap.add_argument("--trials", type=int, default=4000)
    args = ap.parse_args()
    rnd = random.Random(1)
    # which quad a shading lane sits on: floor / ceiling pieces / walls / block faces are hit about equally often per quad in the Cornell box,
    # the light and the small ceiling strips rarely: 14 of the 19 quads carry the weight (a flat choice among them; the conclusion does not hinge on it)
    hit_quads = list(range(14))
    def pick_quads(active):
        return [rnd.choice(hit_quads) if rnd.random() < active else None for _ in range(64)]
    rows = []
    for stride_words in (40, 44):
        tot = {}
        for _ in range(args.trials):
            q = pick_quads(0.96)                                  # lanes shading a hit: 61.4 of 64
            # the record's fields a shading lane reads (csrc/ssx_kernels.hip path_step): normal[which] (12 B at word 20 + 3 which), the four material
            # words (16 B at word 26), the albedo table descriptor (16 B at word 30); after the trace: albedo_mode (4 B at word 27)
            for name, grp, off, width in (("quad record: normal (b96)", G_B96, 20, 3), ("quad record: material words (b128)", G_B128, 26, 4),
                                          ("quad record: albedo descriptor (b128)", G_B128, 30, 4), ("quad record: albedo_mode after the trace (b32)", G_B32, 27, 1)):
                addrs = [None if x is None else 4 * (x * stride_words + off + (3 * rnd.randint(0, 1) if width == 3 else 0)) for x in q]
                b, e = cycles(grp, addrs, width)
                t = tot.setdefault(name, [0, 0]); t[0] += b; t[1] += e
        rows.append((stride_words, tot))
    print("quad records (per wave iteration, mean over %d trials): base LDS cycles, conflict cycles" % args.trials)
    for stride, tot in rows:
        sb = sum(v[0] for v in tot.values()) / args.trials; se = sum(v[1] for v in tot.values()) / args.trials
        print("  stride %d words: base %.1f, conflicts %.1f" % (stride, sb, se))
        for k, v in tot.items():
            print("      %-52s base %5.1f  conflicts %5.1f" % (k, v[0] / args.trials, v[1] / args.trials))
    d = (sum(v[1] for v in rows[0][1].values()) - sum(v[1] for v in rows[1][1].values())) / args.trials
    print("  model: stride 40 -> 44 removes %.1f conflict cycles per iteration; measured (SQ_LDS_BANK_CONFLICT 9.634e8 -> 8.619e8 over 4.70 M iterations): %.1f" % (d, (9.634e8 - 8.619e8) / 4.70e6))

    # pass 2: a candidate's offset record (8 B, table of 38) and its three vertex records (3 x 12 B of 16-byte records, 28 vertices x 3 axis permutations)
    tot = {}
    for occupancy, trips, label in ((20.1 / 64, 4.34, "primary"), (29.1 / 64, 2.15, "shadow")):
        for _ in range(args.trials):
            tri = [rnd.randrange(38) if rnd.random() < occupancy else None for _ in range(64)]
            perm = [rnd.randrange(3) for _ in range(64)]
            b, e = cycles(G_B64, [None if t is None else 8 * t for t in tri], 2)
            t = tot.setdefault("pass 2 %s: offset record (b64)" % label, [0, 0, trips]); t[0] += b; t[1] += e
            for v in range(3):
                addrs = [None if t is None else 16 * (28 * perm[l] + rnd.randrange(28)) for l, t in enumerate(tri)]
                b, e = cycles(G_B96, addrs, 3)
                t2 = tot.setdefault("pass 2 %s: three vertex records (b96)" % label, [0, 0, trips]); t2[0] += b; t2[1] += e
    print("pass 2 (per trip; x trips per iteration)")
    p2 = 0.0
    for k, v in tot.items():
        print("      %-52s base %5.1f  conflicts %5.2f  x %.2f trips = %5.1f conflict cycles per iteration" % (k, v[0] / args.trials, v[1] / args.trials, v[2], v[1] / args.trials * v[2]))
        p2 += v[1] / args.trials * v[2]
    # spectrum gathers: a lane reads data[c], data[c + 1] of ITS table (ds_read2_b32: two b32 accesses), c spread over a quarter of the table per hero wavelength
    tot = [0, 0]
    n_tables = 3   # constant-albedo tables in play within one wave (white, light, block): lanes on different tables at different offsets
    bases = [rnd.randrange(0, 4000) for _ in range(n_tables)]
    for _ in range(args.trials):
        for hero in range(4):
            for d in (0, 1):
                addrs = [4 * (bases[rnd.randrange(n_tables)] + hero * 15 + rnd.randrange(16) + d) if rnd.random() < 0.8 else None for _ in range(64)]
                b, e = cycles(G_B32, addrs, 1)
                tot[0] += b; tot[1] += e
    print("albedo gather (4 hero wavelengths x ds_read2_b32, lanes on %d tables): base %.1f, conflicts %.1f per iteration" % (n_tables, tot[0] / args.trials, tot[1] / args.trials))
    print("sum of the modelled sites: %.0f conflict cycles per iteration (quad records %.0f, pass 2 %.0f, albedo gather %.0f); measured total: %.0f" % (
        sum(v[1] for v in rows[0][1].values()) / args.trials + p2 + tot[1] / args.trials, sum(v[1] for v in rows[0][1].values()) / args.trials, p2, tot[1] / args.trials, 9.634e8 / 4.70e6))


if __name__ == "__main__":
    main()
