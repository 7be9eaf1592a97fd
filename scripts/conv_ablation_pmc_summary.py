import csv, glob, collections, sys
for d in sorted(glob.glob('gpurun_out/abl_pmc/*')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'conv3x3_bf16_kernel' in r['Kernel_Name']:
                agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + '/*/*kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            if 'conv3x3_bf16_kernel' in r['Kernel_Name']:
                dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, c in agg.items():
        n = len(c['GRBM_GUI_ACTIVE']); m = lambda x: sum(c[x]) / max(len(c[x]), 1)
        cyc = m('GRBM_GUI_ACTIVE') / 8
        us = sum(dur[k]) / len(dur[k]) / 1e3
        print(f"{d.split('/')[-1]:10s} {k[22:60]:40s} n={n:3d} {us:7.1f} us  clk {cyc/us/1e3:5.2f} GHz  mfma_busy {m('SQ_VALU_MFMA_BUSY_CYCLES')/(cyc*1024):.3f}  wait_inst/wave {m('SQ_WAIT_INST_ANY')/m('SQ_WAVE_CYCLES'):.3f}  wait_lds/wave {m('SQ_WAIT_INST_LDS')/m('SQ_WAVE_CYCLES'):.3f}")
