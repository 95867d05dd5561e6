"""Generates omr-img-corrector_amd/csrc/slane_asm.inc: the scan-lane sweep's wave program as
gfx950 assembly text, one variant per segment-slot class (S = 2 / 4 / 8 slots per destination word).

Why assembly: the wave keeps its source words in a ring of 65 VGPRs addressed through the gfx9 VGPR index mode (M0),
its segment descriptors (mask, ring index | shift) live in SGPRs filled by s_load_dwordx8/x16, and the column
counters are a carry-save tree in fixed registers -- none of which HIP C++ can express (no dynamically indexed
register arrays, no SGPR arrays).  tools/slane_mb.hip / slane_mb2.hip measured the pieces on the hardware.

Register map (fixed; the statement clobbers s0-s13, s16-s101, v4-v124):
  s[0:1] segment stream  s[2:3] fetch stream  s[4:5] row-count row  s[6:7] counter dump  s[8:11] bit-image descriptor
  s12 rows left  s13 row-count pitch  s[16:19] / s[20:23] fetch dwords of the even / odd row  s[24:27] fetches to commit
  s28 row index  s29 / s33 offsets  s30 / s31 shifts  s32 segments of the word  s[36:67] / s[68:99] segments even / odd
  v[4:21] / v[22:39] column counters of word 0 / 1: planes p0..p11, pending carries c0..c5
  v40-v43 carries  v44 scratch  v45 row count  v46 / v47 odd row's words  v[48:55] aligned windows  v[56:59] fetched
  entries  v[60:124] ring (register 124 = dummy)
Usage: python tools/gen_slane_asm.py   (writes the .inc; the build only reads it)"""
import os

RING = 60
T0 = 56
W0 = 48
DODD = (46, 47)
CNT = 45
TMP = 44
CARRY = ((40, 41), (42, 43))   # per word: tA, tB
P = (4, 22)                    # planes p0..p11 of word k
ST = (16, 34)                  # pending carries c0..c5 of word k
FA, FB, FP = 16, 20, 24
SA, SB = 36, 68


def word(out, k, sbase, S, dreg, tag):
    """segments of word k from the SGPR set at sbase -> VGPR dreg"""
    G = {2: 2, 4: 2, 8: 4}[S]
    m = lambda j: "s%d" % (sbase + k * 2 * S + 2 * j)
    p = lambda j: "s%d" % (sbase + k * 2 * S + 2 * j + 1)
    if S > G:
        out.append("s_bfe_u32 s32, %s, 0x50010" % p(0))
    for g in range(S // G):
        js = list(range(g * G, (g + 1) * G))
        out.append("s_set_gpr_idx_on %s, gpr_idx(SRC0,SRC1)" % p(js[0]))
        for n, j in enumerate(js):
            sh = "s%d" % (30 + (n & 1))
            if n:
                out.append("s_set_gpr_idx_idx %s" % p(j))
            out.append("s_lshr_b32 %s, %s, 8" % (sh, p(j)))
            out.append("v_alignbit_b32 v%d, v%d, v%d, %s" % (W0 + n, RING + 1, RING, sh))
        out.append("s_set_gpr_idx_off")
        for n, j in enumerate(js):
            if g == 0 and n == 0:
                out.append("v_and_b32 v%d, %s, v%d" % (dreg, m(j), W0 + n))
            else:
                out.append("v_and_or_b32 v%d, v%d, %s, v%d" % (dreg, W0 + n, m(j), dreg))
        if g + 1 < S // G:
            out.append("s_cmp_le_u32 s32, %d" % ((g + 1) * G))
            out.append("s_cbranch_scc1 %s" % tag)
    if S > G:
        out.append("%s:" % tag)


def commit_and_fetch(out, fset):
    out.append("s_waitcnt vmcnt(0)")
    out.append("s_set_gpr_idx_on s%d, gpr_idx(DST)" % FP)
    for f in range(4):
        if f:
            out.append("s_set_gpr_idx_idx s%d" % (FP + f))
        out.append("v_mov_b32 v%d, v%d" % (RING, T0 + f))
    out.append("s_set_gpr_idx_off")
    for f in range(4):
        so = "s%d" % (29 if f % 2 == 0 else 33)
        out.append("s_andn2_b32 %s, s%d, 0xff" % (so, fset + f))
        out.append("buffer_load_dword v%d, %%[lane4], s[8:11], %s offen" % (T0 + f, so))
    out.append("s_mov_b64 s[%d:%d], s[%d:%d]" % (FP, FP + 1, fset, fset + 1))
    out.append("s_mov_b64 s[%d:%d], s[%d:%d]" % (FP + 2, FP + 3, fset + 2, fset + 3))


def row_count(out, d0, d1):
    out.append("v_bcnt_u32_b32 v%d, v%d, 0" % (CNT, d0))
    out.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (CNT, d1, CNT))
    out.append("global_atomic_add %%[lane4], v%d, s[4:5]" % CNT)
    out.append("s_add_u32 s4, s4, s13")
    out.append("s_addc_u32 s5, s5, 0")


def maj(out, d, a, b, c):
    out.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0xe8" % (d, a, b, c))


def xor3(out, d, a, b, c):
    out.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96" % (d, a, b, c))


def carry_save(out, L):
    """odd row: (even row's word in c0, this row's word in DODD) -> the carry-save tree; bits 1.. of the row index
    say how far the carry travels"""
    # level 0 always
    for k in range(2):
        maj(out, CARRY[k][0], P[k], ST[k], DODD[k])
        xor3(out, P[k], P[k], ST[k], DODD[k])
    cur = 0  # carry sits in CARRY[k][cur]
    for lv in range(1, 6):
        out.append("s_bitcmp1_b32 s28, %d" % lv)
        out.append("s_cbranch_scc1 %s_add%d" % (L, lv))
        for k in range(2):
            out.append("v_mov_b32 v%d, v%d" % (ST[k] + lv, CARRY[k][cur]))
        out.append("s_branch %s_done" % L)
        out.append("%s_add%d:" % (L, lv))
        for k in range(2):
            maj(out, CARRY[k][cur ^ 1], P[k] + lv, ST[k] + lv, CARRY[k][cur])
            xor3(out, P[k] + lv, P[k] + lv, ST[k] + lv, CARRY[k][cur])
        cur ^= 1
    # the carry out of level 5 ripples into planes 6..11 (once per 64 rows)
    for k in range(2):
        c = cur
        for lv in range(6, 12):
            out.append("v_and_b32 v%d, v%d, v%d" % (CARRY[k][c ^ 1], P[k] + lv, CARRY[k][c]))
            out.append("v_xor_b32 v%d, v%d, v%d" % (P[k] + lv, P[k] + lv, CARRY[k][c]))
            c ^= 1
    out.append("%s_done:" % L)


def seg_load(out, sbase, S, off_rows):
    nd = 4 * S  # dwords per row: 2 words x S x 2
    byte = off_rows * nd * 4
    if nd == 8:
        out.append("s_load_dwordx8 s[%d:%d], s[0:1], %d" % (sbase, sbase + 7, byte))
    elif nd == 16:
        out.append("s_load_dwordx16 s[%d:%d], s[0:1], %d" % (sbase, sbase + 15, byte))
    else:
        out.append("s_load_dwordx16 s[%d:%d], s[0:1], %d" % (sbase, sbase + 15, byte))
        out.append("s_load_dwordx16 s[%d:%d], s[0:1], %d" % (sbase + 16, sbase + 31, byte + 64))


def kernel(S):
    o = []
    L = "%=" # unique label suffix per asm statement
    o += ["s_load_dwordx8 s[0:7], %[desc], 0", "s_load_dwordx4 s[8:11], %[desc], 32", "s_load_dwordx2 s[12:13], %[desc], 48"]
    for v in list(range(4, 48)) + list(range(T0, RING + 65)):
        o.append("v_mov_b32 v%d, 0" % v)
    o.append("s_mov_b32 s28, 0")
    for f in range(4):
        o.append("s_mov_b32 s%d, 64" % (FP + f))
    o.append("s_waitcnt lgkmcnt(0)")
    o.append("s_load_dwordx4 s[%d:%d], s[2:3], 0" % (FA, FA + 3))
    seg_load(o, SA, S, 0)
    o.append("s_waitcnt lgkmcnt(0)")
    o.append("L%s_loop:" % L)
    # ---- even row
    o.append("s_load_dwordx4 s[%d:%d], s[2:3], 16" % (FB, FB + 3))
    seg_load(o, SB, S, 1)
    commit_and_fetch(o, FA)
    word(o, 0, SA, S, ST[0], "L%s_e0" % L)
    word(o, 1, SA, S, ST[1], "L%s_e1" % L)
    row_count(o, ST[0], ST[1])
    o.append("s_waitcnt lgkmcnt(0)")
    # ---- odd row
    o.append("s_load_dwordx4 s[%d:%d], s[2:3], 32" % (FA, FA + 3))
    seg_load(o, SA, S, 2)
    commit_and_fetch(o, FB)
    word(o, 0, SB, S, DODD[0], "L%s_o0" % L)
    word(o, 1, SB, S, DODD[1], "L%s_o1" % L)
    row_count(o, DODD[0], DODD[1])
    carry_save(o, "L%s_cs" % L)
    o += ["s_add_u32 s28, s28, 2",
          "s_add_u32 s0, s0, %d" % (2 * 4 * S * 4), "s_addc_u32 s1, s1, 0",
          "s_add_u32 s2, s2, 32", "s_addc_u32 s3, s3, 0",
          "s_sub_u32 s12, s12, 2", "s_waitcnt lgkmcnt(0)", "s_cmp_lg_u32 s12, 0", "s_cbranch_scc1 L%s_loop" % L]
    # ---- dump the 36 counter registers: [word][p0..p11, c0..c5][lane]
    o.append("s_waitcnt vmcnt(0)")
    for i in range(36):
        if i and i % 16 == 0:
            o += ["s_add_u32 s6, s6, 4096", "s_addc_u32 s7, s7, 0"]
        o.append("global_store_dword %%[lane4], v%d, s[6:7] offset:%d" % (4 + i, (i % 16) * 256))
    o.append("s_waitcnt vmcnt(0)")
    return o


CLOB = ['"memory"', '"scc"', '"vcc"', '"m0"'] + ['"s%d"' % i for i in list(range(0, 14)) + list(range(16, 102))] + \
       ['"v%d"' % i for i in range(4, 125)]

out = ["// GENERATED by tools/gen_slane_asm.py -- do not edit; see that file for the register map.\n"]
for S in (2, 4, 8):
    body = "\\n\\t\"\n    \"".join(kernel(S))
    out.append("#define SLANE_ASM_S%d \\\n    \"%s\\n\\t\"\n" % (S, body.replace("\n", " \\\n")))
out.append("#define SLANE_ASM_CLOBBERS %s\n" % ", ".join(CLOB))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omr-img-corrector_amd", "csrc", "slane_asm.inc")
open(path, "w").write("".join(out))
print("wrote", path, {S: len(kernel(S)) for S in (2, 4, 8)}, "instructions")
