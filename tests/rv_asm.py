"""A few dozen lines of rv64im assembler and an ELF64 writer, for executor tests that need programs the reference's guest
binaries do not contain (division edge cases, x0 destinations, misaligned accesses). Encodings: the RISC-V unprivileged
specification, chapter "RV32/64G Instruction Set Listings"."""
import struct

R_OPS = {  # name: (opcode, funct3, funct7)
    "add": (0x33, 0, 0x00), "sub": (0x33, 0, 0x20), "sll": (0x33, 1, 0x00), "slt": (0x33, 2, 0x00), "sltu": (0x33, 3, 0x00),
    "xor": (0x33, 4, 0x00), "srl": (0x33, 5, 0x00), "sra": (0x33, 5, 0x20), "or": (0x33, 6, 0x00), "and": (0x33, 7, 0x00),
    "mul": (0x33, 0, 0x01), "mulh": (0x33, 1, 0x01), "mulhsu": (0x33, 2, 0x01), "mulhu": (0x33, 3, 0x01), "div": (0x33, 4, 0x01),
    "divu": (0x33, 5, 0x01), "rem": (0x33, 6, 0x01), "remu": (0x33, 7, 0x01),
    "addw": (0x3B, 0, 0x00), "subw": (0x3B, 0, 0x20), "sllw": (0x3B, 1, 0x00), "srlw": (0x3B, 5, 0x00), "sraw": (0x3B, 5, 0x20),
    "mulw": (0x3B, 0, 0x01), "divw": (0x3B, 4, 0x01), "divuw": (0x3B, 5, 0x01), "remw": (0x3B, 6, 0x01), "remuw": (0x3B, 7, 0x01),
}
I_OPS = {"addi": (0x13, 0), "slti": (0x13, 2), "sltiu": (0x13, 3), "xori": (0x13, 4), "ori": (0x13, 6), "andi": (0x13, 7), "addiw": (0x1B, 0),
         "lb": (0x03, 0), "lh": (0x03, 1), "lw": (0x03, 2), "ld": (0x03, 3), "lbu": (0x03, 4), "lhu": (0x03, 5), "lwu": (0x03, 6), "jalr": (0x67, 0)}
SHIFT_OPS = {"slli": (0x13, 1, 0x00), "srli": (0x13, 5, 0x00), "srai": (0x13, 5, 0x10), "slliw": (0x1B, 1, 0x00), "srliw": (0x1B, 5, 0x00),
             "sraiw": (0x1B, 5, 0x10)}
S_OPS = {"sb": 0, "sh": 1, "sw": 2, "sd": 3}
B_OPS = {"beq": 0, "bne": 1, "blt": 4, "bge": 5, "bltu": 6, "bgeu": 7}


def enc(name, *a):
    """One instruction word. R: (rd, rs1, rs2); I / loads / jalr: (rd, rs1, imm); shifts: (rd, rs1, shamt); stores: (rs2, rs1, imm);
    branches: (rs1, rs2, offset); lui / auipc: (rd, imm20 << 12); jal: (rd, offset); ecall: ()."""
    if name in R_OPS:
        op, f3, f7 = R_OPS[name]
        rd, rs1, rs2 = a
        return f7 << 25 | rs2 << 20 | rs1 << 15 | f3 << 12 | rd << 7 | op
    if name in I_OPS:
        op, f3 = I_OPS[name]
        rd, rs1, imm = a
        return (imm & 0xFFF) << 20 | rs1 << 15 | f3 << 12 | rd << 7 | op
    if name in SHIFT_OPS:
        op, f3, hi = SHIFT_OPS[name]
        rd, rs1, sh = a
        return hi << 26 | sh << 20 | rs1 << 15 | f3 << 12 | rd << 7 | op
    if name in S_OPS:
        rs2, rs1, imm = a
        return ((imm >> 5) & 0x7F) << 25 | rs2 << 20 | rs1 << 15 | S_OPS[name] << 12 | (imm & 0x1F) << 7 | 0x23
    if name in B_OPS:
        rs1, rs2, off = a
        return (((off >> 12) & 1) << 31 | ((off >> 5) & 0x3F) << 25 | rs2 << 20 | rs1 << 15 | B_OPS[name] << 12 | ((off >> 1) & 0xF) << 8
                | ((off >> 11) & 1) << 7 | 0x63)
    if name in ("lui", "auipc"):
        rd, imm = a
        return (imm & 0xFFFFF000) | rd << 7 | (0x37 if name == "lui" else 0x17)
    if name == "jal":
        rd, off = a
        return ((off >> 20) & 1) << 31 | ((off >> 1) & 0x3FF) << 21 | ((off >> 11) & 1) << 20 | ((off >> 12) & 0xFF) << 12 | rd << 7 | 0x6F
    if name == "ecall":
        return 0x73
    raise KeyError(name)


def li(rd, value):
    """Instructions loading the 64-bit constant `value` into rd: lui / addiw for 32-bit values, else the upper half that way and
    the lower half in 11 + 11 + 10-bit pieces (slli / ori)."""
    value &= (1 << 64) - 1
    sv = value - (1 << 64) if value >> 63 else value
    if -2048 <= sv < 2048:
        return [enc("addi", rd, 0, sv)]
    if -(1 << 31) <= sv < (1 << 31):
        lo = ((sv & 0xFFF) ^ 0x800) - 0x800
        hi = (sv - lo) & 0xFFFFFFFF
        return [enc("lui", rd, hi)] + ([enc("addiw", rd, rd, lo)] if lo else [])
    out, lo = li(rd, sv >> 32), value & 0xFFFFFFFF
    for shift, piece in ((11, lo >> 21), (11, (lo >> 10) & 0x7FF), (10, lo & 0x3FF)):
        out.append(enc("slli", rd, rd, shift))
        if piece:
            out.append(enc("ori", rd, rd, piece))
    return out


def halt(code=0):
    """`li a0, code; li t0, 0 (HALT); ecall`."""
    return li(10, code) + li(5, 0) + [enc("ecall")]


def elf(words, data=b"", base=0x78000000, data_addr=0x78100000):
    """ELF64 executable: one R+X segment with `words` at `base` (entry = base), optionally one R+W segment with `data`."""
    text = b"".join(struct.pack("<I", w & 0xFFFFFFFF) for w in words)
    segs = [(5, base, text)] + ([(6, data_addr, data)] if data else [])
    ehsize, phsize = 64, 56
    off = ehsize + phsize * len(segs)
    off = (off + 15) & ~15
    ph, body = b"", b""
    for flags, vaddr, blob in segs:
        ph += struct.pack("<IIQQQQQQ", 1, flags, off + len(body), vaddr, vaddr, len(blob), len(blob), 0x1000)
        body += blob + b"\0" * (-len(blob) % 16)
    eh = b"\x7fELF" + bytes([2, 1, 1, 0]) + b"\0" * 8 + struct.pack("<HHIQQQIHHHHHH", 2, 243, 1, base, ehsize, 0, 0, ehsize, phsize, len(segs), 64, 0, 0)
    pad = b"\0" * (off - ehsize - len(ph))
    return eh + ph + pad + body
