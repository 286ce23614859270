#!/usr/bin/env python3
"""Writes gripnet_amd/csrc/rgcn_pair_asm.inc: the gather of one unit of the destination-major relational kernel as one
block of gfx950 assembly (string literals GN_PAIR_UNIT_ASM_B64 / _B32 for two bases / one base per lane).

A unit = eight sections; a section = the edges of four (destination, source) pairs (one per 16-lane group) in lock step,
`count` blocks of four edges each, summed into the section's accumulator P_t.  The code is a software pipeline over
the unit's blocks, flat across sections:

  position consuming block b:  wait for rows(b)          s_waitcnt lgkmcnt(5)   (younger: word(b+3), rows(b+1) x 4)
                               P_t += rows(b)            three adds inside the buffer, one into P_t
                               addresses of rows(b+2)    from word(b+2): it sits before rows(b) in the queue, so it is here
                               request word(b+4)         window of the stream in LDS; refill by LDS-DMA at quarter crossings
                               request rows(b+2) x 4

Buffers and word registers rotate with period three, so the body of a section is three positions long and is entered
at the position its first block falls on; which of the three exits a section leaves by decides the entry of the next:
all static labels, no per-block bookkeeping of the rotation.  The word of a block goes round each lane quad (DPP): lane p
reads the rows of edges p, p+1, p+2, p+3 (mod 4) of its group's block.  The units of a wave follow each other in the stream
without a gap: the tail of a unit requests rows(N), rows(N+1) - the first two blocks of the next unit - and drops them, and
the next unit re-reads its first four words from the window (soff stays on its word 4).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "gripnet_amd", "csrc", "rgcn_pair_asm.inc")

DPP = ["", " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf", " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf",
       " quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf"]


RB = 92            # first of the 28 VGPRs the block owns (v92-v119: the kernel runs four waves per SIMD, 128 registers)
NREGS = 28
W = ["v{}".format(RB + 24 + k) for k in range(3)]
WADDR = VDMA = "v{}".format(RB + 27)   # the refill's address is consumed when it issues, in front of the word's address
# (a row's LDS address is computed in the first register of the buffer row it is read into: consumed when the read issues)


def regs(bt):
    """buffers B0..B2 (4 rows each), words W0..W2"""
    if bt == 2:
        buf = [["v[{}:{}]".format(RB + 8 * k + 2 * i, RB + 1 + 8 * k + 2 * i) for i in range(4)] for k in range(3)]
    else:
        buf = [["v{}".format(RB + 4 * k + i) for i in range(4)] for k in range(3)]
    return buf, W


def issue(lines, rd, use, req, into, behind=0):
    """rows of the block whose word is `use` into `into`; word `req` from the window: the next one (soff, which advances),
    or - `behind` > 0, a unit's prologue - the one `behind` bytes back (already passed by the unit before)"""
    A = [int(r.replace("v[", "").replace("v", "").split(":")[0]) for r in into]
    lines.append("v_add_u32 v{}, {}, %[lo]".format(A[0], use))
    for j in (1, 2, 3):
        lines.append("v_add_u32_dpp v{}, {}, %[lo]{}".format(A[j], use, DPP[j]))
    if behind:
        lines += ["s_sub_u32 s94, %[soff], {}".format(behind), "s_and_b32 s94, s94, 0x7ff", "v_add_u32 {}, s94, %[rlane]".format(WADDR),
                  "ds_read_b32 {}, {}".format(req, WADDR)]
        for j in range(4):
            lines.append("{} {}, v{}".format(rd, into[j], A[j]))
        return
    lines += [
        "s_and_b32 s94, %[soff], 0x1c0",
        "s_cbranch_scc1 1f",
        # entering a quarter (eight blocks, 512 bytes) of the 2 KB window: its refill is the second youngest vector-memory
        # request; the quarter two behind is dead (the words in flight are at most four blocks back): refill it with the
        # blocks two quarters ahead - one LDS-DMA instruction of the lower 32 lanes (16 bytes each)
        "s_waitcnt vmcnt(1)",
        "s_sub_u32 s94, %[soff], 0x400",
        "s_and_b32 s94, s94, 0x600",
        "s_add_u32 m0, s94, %[rbase]",
        "v_lshl_add_u32 {}, %[l4], 2, %[sdma]".format(VDMA),                 # lane * 16 (this lane's bytes of the 1 KB refill) + offset
        "s_add_u32 %[sdma], %[sdma], 0x200",
        "s_mov_b32 exec_hi, 0",
        "global_load_lds_dwordx4 {}, %[sbase]".format(VDMA),
        "s_mov_b32 exec_hi, -1",
        "1:",
        "v_add_u32 {}, %[soff], %[rlane]".format(WADDR),
        "s_add_u32 %[soff], %[soff], 64",
        "s_and_b32 %[soff], %[soff], 0x7ff",
        "ds_read_b32 {}, {}".format(req, WADDR),
    ]
    for j in range(4):
        lines.append("{} {}, v{}".format(rd, into[j], A[j]))


def unit(bt):
    rd, add = ("ds_read_b64", "v_pk_add_f32") if bt == 2 else ("ds_read_b32", "v_add_f32")
    buf, w = regs(bt)
    L = []
    L.append("s_mov_b32 s92, m0")
    for t in range(8):
        L.append(("v_mov_b64 %[p{}], 0" if bt == 2 else "v_mov_b32 %[p{}], 0").format(t))
    # the units of a wave follow each other in the stream without a gap: this unit's words 0..3 were already requested (and
    # its rows 0 and 1, dropped) by the tail of the unit before - soff stands at word 4 - so they are read again from the window
    for k, back in ((0, 256), (1, 192)):
        L += ["s_sub_u32 s94, %[soff], {}".format(back), "s_and_b32 s94, s94, 0x7ff", "v_add_u32 {}, s94, %[rlane]".format(WADDR),
              "ds_read_b32 {}, {}".format(w[k], WADDR)]
    L.append("s_waitcnt lgkmcnt(0)")
    issue(L, rd, w[0], w[2], buf[0], behind=128)      # rows(0), word(2)
    issue(L, rd, w[1], w[0], buf[1], behind=64)       # rows(1), word(3)
    L += ["s_bfe_u32 s93, %[c03], 0x80000", "s_sub_u32 s93, 0, s93"]
    for t in range(8):
        for ph in range(3):
            L.append("S{}_{}_%=:".format(t, ph))
            L.append("s_waitcnt lgkmcnt(5)")
            b = buf[ph]
            L += ["{} {}, {}, {}".format(add, b[0], b[0], b[1]), "{} {}, {}, {}".format(add, b[2], b[2], b[3]),
                  "{} {}, {}, {}".format(add, b[0], b[0], b[2]), "{} %[p{}], %[p{}], {}".format(add, t, t, b[0])]
            issue(L, rd, w[(ph + 2) % 3], w[(ph + 1) % 3], buf[(ph + 2) % 3])
            L += ["s_add_u32 s93, s93, 1", "s_cbranch_scc1 E{}_{}_%=".format(t, ph)]
        L.append("s_branch S{}_0_%=".format(t))
        for ph in range(3):
            L.append("E{}_{}_%=:".format(t, ph))
            if t == 7:
                L.append("s_branch D_%=")
            else:
                nt = t + 1
                L += ["s_bfe_u32 s93, %[c{}], {}".format("03" if nt < 4 else "47", hex(0x80000 | 8 * (nt % 4))),
                      "s_sub_u32 s93, 0, s93", "s_branch S{}_{}_%=".format(nt, (ph + 1) % 3)]
    L += ["D_%=:", "s_waitcnt lgkmcnt(0)", "s_mov_b32 m0, s92"]
    return L


def render():
    out = ["// Generated by tools/gen_pair_asm.py - do not edit.  The gather of one unit (see rgcn_pair.hip).\n",
           "#define GN_PAIR_ASM_CLOBBERS {}\n\n".format(", ".join('"v{}"'.format(RB + i) for i in range(NREGS)))]
    for bt, name in ((2, "GN_PAIR_UNIT_ASM_B64"), (1, "GN_PAIR_UNIT_ASM_B32")):
        out.append("#define {} \\\n".format(name))
        lines = unit(bt)
        for i, ln in enumerate(lines):
            sep = "\\n" if ln.endswith(":") else "\\n\\t"
            out.append('    "{}{}"{}\n'.format(ln, sep, " \\" if i + 1 < len(lines) else ""))
        out.append("\n")
    return "".join(out)


def main():
    import argparse
    ap = argparse.ArgumentParser(description="Generates gripnet_amd/csrc/rgcn_pair_asm.inc.  Without --write nothing is written: "
                                             "the generated text is compared with the tracked file (exit code 1 if they differ).")
    ap.add_argument("--write", action="store_true", help="overwrite the tracked .inc (only if the text changed)")
    ap.add_argument("--stdout", action="store_true", help="print the generated text instead")
    args = ap.parse_args()
    text = render()
    if args.stdout:
        sys.stdout.write(text)
        return 0
    have = open(OUT).read() if os.path.exists(OUT) else None
    if have == text:
        print("up to date:", OUT)
        return 0
    if not args.write:
        print("differs from the generated text (run with --write):", OUT)
        return 1
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
