"""Instruction counts and register use of one kernel in a hipcc -save-temps .s file:  isa_stats.py file.s substring"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if key in n]
for n in names:
    i = s.index('\n' + n + ':')
    j = s.index('.Lfunc_end', i)
    body = s[i:j]
    print(n, 'lines', body.count('\n'))
    for pat in ['v_mfma', 'v_pk_add_f16', 'v_pk_', 'ds_read_b128', 'ds_read', 'ds_write_b128', 'ds_write', 'scratch_', 's_barrier', 'v_cvt_pk_f16', 'v_cvt_f16_f32',
                'global_load_lds', 'global_store', 'v_permlane', 'v_mov_b32', 's_waitcnt', 's_nop', 'v_accvgpr', '_dpp']:
        print('  %-16s %d' % (pat, len(re.findall(pat, body))))
    k = s.find('.name:', s.find('amdhsa.kernels'))
    m = re.search(r'\.agpr_count:\s+(\d+).*?\.name:\s+' + re.escape(n) + r'.*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', s, re.S)
    blk = s[s.rfind('- .agpr_count', 0, s.index('.name:           ' + n)):]
    blk = blk[:blk.index('.wavefront_size')]
    print('  ' + ' '.join(x.strip() for x in re.findall(r'\.(?:agpr_count|vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+\d+', blk)))
