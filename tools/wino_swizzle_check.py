"""LDS layout of trunkw_kernel's raw rows (halo tiles and the intermediate ring): is every T1 fragment read
(ds_read_b128, lane (o, p) reads octet 4ch+o of column 2p+k) and every epilogue ring write (ds_write_b128, 8
contiguous lanes = 8 consecutive pairs, one column parity, one octet) bank-conflict free?

Layout: a row is 34 pixel records of 128 bytes; column cc = 2h + par sits at record h + 17*par (even columns
first, then the odd ones) and channel octet `oct` of it in 16-byte slot oct ^ (h & 7).
Lane groups and banking: MI355X_MICROARCH.md, LDS table."""
RG = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG += [[l + 32 for l in g] for g in RG]

def unit(cc, octet):
    h, par = cc >> 1, cc & 1
    return (h + 17 * par) * 8 + (octet ^ (h & 7))

def reads_ok():
    for k in range(4):
        for ch in range(2):
            for g in RG:
                banks = {unit(2 * (l & 15) + k, 4 * ch + (l >> 4)) % 16 for l in g}     # 64 banks x 4 B = 16 units
                if len(banks) != 16:
                    return False, (k, ch, g)
    return True, None

def writes_ok():
    # lane (cg, p): column 2p + (cg & 1), octet 2m + (cg >> 1); ds_write_b128 groups: 8 contiguous lanes, 32 banks = 8 units
    for m in range(4):
        for cg in range(4):
            for p0 in (0, 8):
                banks = {unit(2 * p + (cg & 1), 2 * m + (cg >> 1)) % 8 for p in range(p0, p0 + 8)}
                if len(banks) != 8:
                    return False, (m, cg, p0)
    return True, None

if __name__ == "__main__":
    print("T1 fragment reads conflict-free:", reads_ok())
    print("ring writes conflict-free:", writes_ok())
    us = sorted(unit(cc, o) for cc in range(34) for o in range(8))
    print("row is a permutation of", len(us), "units:", us == list(range(34 * 8)))
