"""Generates omr-img-corrector_amd/csrc/slane_asm.inc: the scan-lane sweep's wave program as
gfx950 assembly text, one variant per segment-slot class (S = 2 / 4 / 8 slots per destination word).

Why assembly: the wave keeps its source words in a ring of 64 VGPRs addressed through the gfx9 VGPR index mode (M0),
its segment descriptors (mask, ring index | shift) live in SGPRs filled by s_load_dwordx8/x16, and the column
counters are a carry-save tree in fixed registers -- none of which HIP C++ can express (no dynamically indexed
register arrays, no SGPR arrays).  tools/slane_mb.hip / slane_mb2.hip measured the pieces on the hardware.

Structure of a turn (four destination rows; row r of the turn uses landing set r, segment set r & 1):
  wait until the loads issued four rows ago have landed (counted: loads return in order, 12 younger ones may fly)
  -> commit them into the ring (v_mov with DST_REL) -> issue this row's four loads -> request the next row's records
  -> two words: per segment v_alignbit (SRC0|SRC1_REL) + v_and_or -> row count (pairs of rows meet the workgroup's
  other strips in LDS, one global atomic per pair row and 16 rows) -> (odd rows) carry-save.
The scalar unit issues one instruction per SIMD every four cycles, like the vector unit: the first version of this loop
(63 scalar against 43 vector instructions per row) was SCALAR-bound.  Hence: index mode stays on for the whole loop
(M0 = 0 = nothing indexed), one s_lshr writes index and mode bits of M0 per segment, fetch offsets and commit
registers come ready-made from the fetch record, the row counts go through a buffer descriptor with a scalar offset.

Register map (fixed; the statement clobbers s0-s13, s16-s100, v1-v127, so lane * 4 arrives in v0):
  s[0:1] segment stream  s[2:3] fetch stream  s[4:7] row-count buffer descriptor  s[8:11] bit-image descriptor
  s12 rows left  s13 row-count pitch  s[16:23] / s[24:31] fetch records even / odd row  s32 row-count offset
  s33 row index  s34 slot class, then segments of the word / scratch  s35 strip's place in its quad  s100 LDS base of the scan group's accumulators  s[36:67] / s[68:99] segment records even / odd row
  v[1:17] / v[18:34] column counters of word 0 / 1: planes p0..p11, pending carries c0..c4
  v35 LDS address of the turn's row-count slots  v36 odd row's count
  v37-v40 carries  v41 row count  v42 / v43 odd row's words  v[44:59] four landing sets of four entries
  v[60:123] ring  v[124:127] aligned windows (v124 doubles as the ring's dummy register 64)
Usage: python tools/gen_slane_asm.py   (writes the .inc; the build only reads it)"""
import os

ABLATE = os.environ.get("SLANE_ABLATE", "").split(",")  # timing probes only (results are wrong): norec, noatomic, nofetch
NOP = []  # ["s_nop 0"]: wait state between a scalar write of M0 and the indexed VALU instruction (probe)
RING = 60
T0 = 44
W0 = 124
DODD = (42, 43)
CNT = 41
CARRY = ((37, 38), (39, 40))   # per word: tA, tB
P = (1, 18)                    # planes p0..p11 of word k
ST = (13, 30)                  # pending carries c0..c4 of word k
NST = 5                        # carry-save levels with a parked carry; the carry out of the last ripples into the planes
LADDR, CNT2 = 35, 36           # LDS address of the turn's row-count slots (lane * 4 + slot offset); odd row's count
NDUMP = 34
FSET = (16, 24)                # fetch record of the even / odd row (8 SGPRs each)
SSET = (36, 68)                # segment record of the even / odd row
AHEAD = 4


def word(out, k, sbase, S, dreg, tag):
    """segments of word k from the SGPR set at sbase -> VGPR dreg.  The wave stays in VGPR index mode for good: M0 = 0
    means "nothing indexed"; per segment ONE scalar instruction (s_lshr m0, pk, 5: ring index + SRC0_REL | SRC1_REL)
    arms the v_alignbit, whose shift operand is pk itself (bits 4:0)."""
    G = {2: 2, 4: 2, 8: 4}[S]
    m = lambda j: "s%d" % (sbase + k * 2 * S + 2 * j)
    p = lambda j: "s%d" % (sbase + k * 2 * S + 2 * j + 1)
    if S > G:
        out.append("s_lshr_b32 s34, %s, 24" % p(0))
    for g in range(S // G):
        js = list(range(g * G, (g + 1) * G))
        for n, j in enumerate(js):
            out.append("s_lshr_b32 m0, %s, 5" % p(j))
            out += NOP
            out.append("v_alignbit_b32 v%d, v%d, v%d, %s" % (W0 + n, RING + 1, RING, p(j)))
        out.append("s_mov_b32 m0, 0")
        out += NOP
        for n, j in enumerate(js):
            if g == 0 and n == 0:
                out.append("v_and_b32 v%d, %s, v%d" % (dreg, m(j), W0 + n))
            else:
                out.append("v_and_or_b32 v%d, v%d, %s, v%d" % (dreg, W0 + n, m(j), dreg))
        if g + 1 < S // G:
            out.append("s_cmp_le_u32 s34, %d" % ((g + 1) * G))
            out.append("s_cbranch_scc1 %s" % tag)
    if S > G:
        out.append("%s:" % tag)


def commit_and_fetch(out, fset, tset):
    """loads return in order: at most (AHEAD - 1) * 4 younger loads may still fly when this row's set has landed
    (row-count atomics in flight only make the wait longer)"""
    t = T0 + 4 * tset
    out.append("s_waitcnt vmcnt(%d)" % ((AHEAD - 1) * 4))
    for f in range(4):
        out.append("s_mov_b32 m0, s%d" % (fset + 4 + f))   # ring register | DST_REL
        out += NOP
        out.append("v_mov_b32 v%d, v%d" % (RING, t + f))
    out.append("s_mov_b32 m0, 0")
    out += NOP
    for f in range(4):
        if "nofetch" not in ABLATE:
            out.append("buffer_load_dword v%d, %%[lane4], s[8:11], s%d offen" % (t + f, fset + f))


def row_count(out, d0, d1, odd, second):
    """Row counts leave the wave as PAIRS: even row in the low half of a dword, odd row in the high half (a count is at
    most 64 per wave and 2480 per row, so neither half can overflow).  The four waves of a workgroup -- four adjacent
    strips of one candidate and scan group -- add their pairs in LDS (ds_add_u32); every 16 rows the workgroup meets
    and each wave sends two of the eight accumulated pair rows on with ONE global atomic per pair row (flush()).
    Without this the kernel issued one global atomic per wave and row: 1.8e9 L2 atomic requests per launch, 25 ms."""
    if not odd:
        out.append("v_bcnt_u32_b32 v%d, v%d, 0" % (CNT, d0))
        out.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (CNT, d1, CNT))
        return
    out.append("v_bcnt_u32_b32 v%d, v%d, 0" % (CNT2, d0))
    out.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (CNT2, d1, CNT2))
    out.append("v_lshl_or_b32 v%d, v%d, 16, v%d" % (CNT, CNT2, CNT))
    if "noatomic" not in ABLATE:
        out.append("ds_add_u32 v%d, v%d offset:%d" % (LADDR, CNT, 256 if second else 0))


def flush(out, L):
    """after rows 12..15 of a block of 16: the workgroup meets, wave w reads pair rows 2w, 2w + 1 of the block's LDS
    buffer, clears them and adds them to the candidate's row counts in memory (s32 = offset of the block's first pair)"""
    out += ["s_and_b32 s34, s33, 12", "s_cmp_lg_u32 s34, 12", "s_cbranch_scc1 %s_nofl" % L]
    if "noatomic" not in ABLATE:
        out += ["s_waitcnt lgkmcnt(0)", "s_barrier",
                "s_and_b32 s34, s33, 16", "s_lshl_b32 s34, s34, 7",        # buffer (bit 4 of the row index) * 2048
                "s_add_u32 s34, s34, s100",                                # + this scan group's accumulators
                "v_add_u32 v%d, s34, %%[lane4]" % CNT2,
                "s_lshl_b32 s34, s35, 9",                                  # + wave * 2 slots * 256 bytes
                "v_add_u32 v%d, s34, v%d" % (CNT2, CNT2),
                "ds_read_b32 v%d, v%d" % (CARRY[0][0], CNT2), "ds_read_b32 v%d, v%d offset:256" % (CARRY[0][1], CNT2),
                "v_mov_b32 v%d, 0" % CARRY[1][0],
                "ds_write_b32 v%d, v%d" % (CNT2, CARRY[1][0]), "ds_write_b32 v%d, v%d offset:256" % (CNT2, CARRY[1][0]),
                "s_mul_i32 s34, s35, s13", "s_lshl_b32 s34, s34, 1", "s_add_u32 s34, s34, s32",
                "s_waitcnt lgkmcnt(0)",
                "buffer_atomic_add v%d, %%[lane4], s[4:7], s34 offen" % CARRY[0][0],
                "s_add_u32 s34, s34, s13",
                "buffer_atomic_add v%d, %%[lane4], s[4:7], s34 offen" % CARRY[0][1]]
    out += ["s_lshl_b32 s34, s13, 3", "s_add_u32 s32, s32, s34", "%s_nofl:" % L]


def maj(out, d, a, b, c):
    out.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0xe8" % (d, a, b, c))


def xor3(out, d, a, b, c):
    out.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96" % (d, a, b, c))


def carry_save(out, L, second):
    """odd row of a pair: (even row's word in c0, this row's word in DODD) -> the carry-save tree.  Row index = s6 + 1
    (first pair of the turn: bit 1 clear, the carry is parked in c1) or s6 + 3 (second pair: bit 1 set, c1 is consumed and
    bits 2.. of s6 say how far the carry travels)."""
    for k in range(2):
        maj(out, CARRY[k][0], P[k], ST[k], DODD[k])
        xor3(out, P[k], P[k], ST[k], DODD[k])
    if not second:
        for k in range(2):
            out.append("v_mov_b32 v%d, v%d" % (ST[k] + 1, CARRY[k][0]))
        return
    cur = 0
    for k in range(2):
        maj(out, CARRY[k][1], P[k] + 1, ST[k] + 1, CARRY[k][0])
        xor3(out, P[k] + 1, P[k] + 1, ST[k] + 1, CARRY[k][0])
    cur = 1
    for lv in range(2, NST):
        out.append("s_bitcmp1_b32 s33, %d" % lv)
        out.append("s_cbranch_scc1 %s_add%d" % (L, lv))
        for k in range(2):
            out.append("v_mov_b32 v%d, v%d" % (ST[k] + lv, CARRY[k][cur]))
        out.append("s_branch %s_done" % L)
        out.append("%s_add%d:" % (L, lv))
        for k in range(2):
            maj(out, CARRY[k][cur ^ 1], P[k] + lv, ST[k] + lv, CARRY[k][cur])
            xor3(out, P[k] + lv, P[k] + lv, ST[k] + lv, CARRY[k][cur])
        cur ^= 1
    for k in range(2):  # the carry out of the last level ripples into the remaining planes (once per 2^NST rows)
        c = cur
        for lv in range(NST, 12):
            out.append("v_and_b32 v%d, v%d, v%d" % (CARRY[k][c ^ 1], P[k] + lv, CARRY[k][c]))
            out.append("v_xor_b32 v%d, v%d, v%d" % (P[k] + lv, P[k] + lv, CARRY[k][c]))
            c ^= 1
    out.append("%s_done:" % L)


def rec_loads(out, which, S, row, force=False):
    """request the records of row `row` of the turn (relative to the stream pointers) into set `which`"""
    if "norec" in ABLATE and not force:
        return
    out.append("s_load_dwordx8 s[%d:%d], s[2:3], %d" % (FSET[which], FSET[which] + 7, row * 32))
    nd = 4 * S
    byte = row * nd * 4
    sb = SSET[which]
    if nd == 8:
        out.append("s_load_dwordx8 s[%d:%d], s[0:1], %d" % (sb, sb + 7, byte))
    elif nd == 16:
        out.append("s_load_dwordx16 s[%d:%d], s[0:1], %d" % (sb, sb + 15, byte))
    else:
        out.append("s_load_dwordx16 s[%d:%d], s[0:1], %d" % (sb, sb + 15, byte))
        out.append("s_load_dwordx16 s[%d:%d], s[0:1], %d" % (sb + 16, sb + 31, byte + 64))


def body(o, S, L):
    """the row loop of one slot class"""
    rec_loads(o, 0, S, 0, True)
    rec_loads(o, 1, S, 1, True)
    o.append("s_waitcnt lgkmcnt(0)")
    o.append("L%s_loop:" % L)
    # LDS address of this turn's two pair slots: buffer = bit 4 of the row index, slot = bits 3:1
    o += ["s_bfe_u32 s34, s33, 0x40001", "s_lshl_b32 s34, s34, 8", "s_add_u32 s34, s34, s100", "v_add_u32 v%d, s34, %%[lane4]" % LADDR]
    for r in range(4):
        w = r & 1
        rec_loads(o, w ^ 1, S, r + 1)       # the next row's records travel while this row is swept
        commit_and_fetch(o, FSET[w], r)
        d = (ST[0], ST[1]) if w == 0 else DODD
        word(o, 0, SSET[w], S, d[0], "L%s_r%dw0" % (L, r))
        word(o, 1, SSET[w], S, d[1], "L%s_r%dw1" % (L, r))
        row_count(o, d[0], d[1], w == 1, r == 3)
        if w:
            carry_save(o, "L%s_cs%d" % (L, r), r == 3)
        if r == 3:
            flush(o, "L%s" % L)
            o += ["s_add_u32 s33, s33, 4",
                  "s_add_u32 s0, s0, %d" % (4 * 4 * S * 4), "s_addc_u32 s1, s1, 0",
                  "s_add_u32 s2, s2, 128", "s_addc_u32 s3, s3, 0",
                  "s_sub_u32 s12, s12, 4"]
        o.append("s_waitcnt lgkmcnt(0)")
    o += ["s_cmp_lg_u32 s12, 0", "s_cbranch_scc1 L%s_loop" % L]


def kernel():
    """ONE statement: descriptor, dispatch on the slot class, the three row loops, the counter dump.  (Three statements
    behind a C++ branch keep the class live across them, and with nearly every register clobbered that costs a 129th
    VGPR -- a whole wave per SIMD.)"""
    o = []
    U = "%="  # unique label suffix per asm statement
    o += ["s_load_dwordx8 s[0:7], %[desc], 0", "s_load_dwordx4 s[8:11], %[desc], 32", "s_load_dwordx2 s[12:13], %[desc], 48",
          "s_load_dword s34, %[desc], 56", "s_load_dword s35, %[desc], 60", "s_load_dword s100, %[desc], 72"]
    for v in range(1, 128):
        o.append("v_mov_b32 v%d, 0" % v)
    o.append("s_waitcnt lgkmcnt(0)")
    o += ["s_mov_b32 s32, 0", "s_mov_b32 s33, 0"]
    o += ["s_set_gpr_idx_on s32, gpr_idx(SRC0)", "s_mov_b32 m0, 0"]  # index mode on for good; M0 = 0: nothing indexed
    o += ["s_cmp_eq_u32 s34, 0", "s_cbranch_scc1 L%s_c0" % U, "s_cmp_eq_u32 s34, 1", "s_cbranch_scc1 L%s_c1" % U]
    for cls, S in ((2, 8), (1, 4), (0, 2)):
        o.append("L%s_c%d:" % (U, cls))
        body(o, S, "%s_c%d" % (U, cls))
        if cls:
            o.append("s_branch L%s_dump" % U)
    # ---- dump the 34 counter registers: [word][p0..p11, c0..c4][lane]
    o.append("L%s_dump:" % U)
    o.append("s_set_gpr_idx_off")
    o.append("s_load_dwordx2 s[32:33], %[desc], 64")
    o.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    for i in range(NDUMP):
        if i and i % 16 == 0:
            o += ["s_add_u32 s32, s32, 4096", "s_addc_u32 s33, s33, 0"]
        o.append("global_store_dword %%[lane4], v%d, s[32:33] offset:%d" % (1 + i, (i % 16) * 256))
    o.append("s_waitcnt vmcnt(0)")
    return o


CLOB = ['"memory"', '"scc"', '"vcc"', '"m0"'] + ['"s%d"' % i for i in list(range(0, 14)) + list(range(16, 101))] + ['"v%d"' % i for i in range(1, 128)]

out = ["// GENERATED by tools/gen_slane_asm.py -- do not edit; see that file for the register map.\n"]
body_txt = "\\n\\t\"\n    \"".join(kernel())
out.append("#define SLANE_ASM \\\n    \"%s\\n\\t\"\n" % body_txt.replace("\n", " \\\n"))
out.append("#define SLANE_ASM_CLOBBERS %s\n" % ", ".join(CLOB))
path = os.environ.get("SLANE_ASM_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omr-img-corrector_amd", "csrc", "slane_asm.inc")
open(path, "w").write("".join(out))
print("wrote", path, len(kernel()), "instructions")
