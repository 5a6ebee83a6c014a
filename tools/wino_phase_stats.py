"""Instruction mix of trunkw_kernel's four phases (general loop bodies) from /tmp/wk/wk.s (tools/wino_isa.sh)."""
import re, collections
src = open('/tmp/wk/wk.s').read()
src = src[src.index('\n_ZN3uva13trunkw'):]
L = src[:src.index('.Lfunc_end')].split('\n')
bar = [i for i, l in enumerate(L) if 's_barrier' in l]
sp0 = [i for i, l in enumerate(L) if 's_setprio 0' in l]
def summary(a, b, name):
    c = collections.Counter()
    for l in L[a:b]:
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if m and not l.strip().startswith(';'): c[m.group(1)] += 1
    valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    salu = sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_waitcnt', 's_nop', 's_barrier'))
    print('%-44s VALU %3d SALU %3d mfma %2d ds_read %2d ds_write %2d gstore %d waitcnt %2d nop %2d' % (
        name, valu, salu, c['v_mfma_f32_16x16x32_f16'], c['ds_read_b128'], c['ds_write_b128'], c['global_store_dwordx4'], c['s_waitcnt'], c['s_nop']))
    print('   ', ' '.join('%s:%d' % (k[2:], v) for k, v in sorted(((k, v) for k, v in c.items() if k.startswith('v_') and 'mfma' not in k), key=lambda kv: -kv[1])[:16]))
# the last three k-loops: B peeled, B general, A general
bk, ak = sp0[-2], sp0[-1]
b_prev = max(b for b in bar if b < bk); b_prev2 = max(b for b in bar if b < b_prev)
summary(b_prev2, b_prev, 'B phase X: epilogue (stores)')
summary(b_prev, min(b for b in bar if b > bk), 'B phase Y: k-loop')
a_b1 = min(b for b in bar if b > ak); a_b0 = max(b for b in bar if b < ak)
summary(a_b0, a_b1, 'A phase X: DMA issue + k-loop')
summary(a_b1, len(L), 'A phase Y: epilogue + raw rows (+ loop tail)')
