import ast,sys
t=open(sys.argv[1]).read().splitlines()
for i,l in enumerate(t):
    if l.startswith('    {'):
        d=ast.literal_eval(l.strip()); name=t[i-1]
        wc=d.get('SQ_WAVE_CYCLES',1); bc=d.get('SQ_BUSY_CYCLES',1)
        print(name[:80])
        print(f"   busy {bc/1e6:6.2f}M waves {d.get('SQ_WAVES',0):7.0f} occ(wave_cyc/busy) {wc/bc:5.1f} VALUactive/busy {d.get('SQ_ACTIVE_INST_VALU',0)/bc:.2f} LDSactive/busy {d.get('SQ_ACTIVE_INST_LDS',0)/bc:.2f} bankconf/busy {d.get('SQ_LDS_BANK_CONFLICT',0)/bc:.2f} any_active/busy {d.get('SQ_ACTIVE_INST_ANY',0)/bc:.2f} wait_inst_any/wc {d.get('SQ_WAIT_INST_ANY',0)/wc:.2f} wait_any/wc {d.get('SQ_WAIT_ANY',0)/wc:.2f} valu/lds {d.get('SQ_INSTS_VALU',0)/max(1,d.get('SQ_INSTS_LDS',1)):.1f}")
