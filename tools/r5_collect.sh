#!/bin/bash
# here, after tools/r5_final.sh ran on the GPU box: condense gpurun_out/ into the tracked profiles/r05_* files
cd "$(dirname "$0")/.."
O=gpurun_out/r5final
for k in 2 3 4 5; do python tools/summarize_profile.py r05_c$k 2>&1 | head -1; done
tail -3 $O/pytest.log > profiles/r05_gputest_summary.txt; tail -2 $O/soak.log >> profiles/r05_gputest_summary.txt
cp $O/latency_host.txt profiles/r05_latency_host.txt
cp $O/bench_default.json profiles/r05_bench_default_line.json
python - <<'PY' > profiles/r05_bench_lines.txt
import json,glob,os
O='gpurun_out/r5final'
print("bench.py lines of tools/r5_final.sh bench (one MI355X); gloo / inprocess lines are functional rehearsals on ONE device;")
print("the counter-derived fields (traffic, lds_frac, valu_issue_frac) are filled from profiles/pmc_traffic.json once the profile passes of the same sources exist")
for f in sorted(glob.glob(O+'/bench_c*.json'))+sorted(glob.glob(O+'/gloo_c*.json'))+sorted(glob.glob(O+'/inproc_c*.json')):
    try: d=json.load(open(f))
    except Exception as e: print(os.path.basename(f),'unreadable',e); continue
    r=d.get('roofline',{})
    line="%-22s %10.3f MFFT/s ms/step %8.3f %-6s" % (os.path.basename(f), d['value']/1e6, d['ms_per_step'], d['scaling'])
    if r: line+=" kern %.3f ms frac %.4f flop %.3f bound %s" % (r['avg_kernel_ms'], r['frac'], r['flop_frac'], r['bound'])
    if 'ranks' in d:
        rk=d['ranks']; line+=" | ranks %d backend %s distinct_devices %d identical %s strong %s" % (rk['world_size'], rk['backend'][:9], rk['distinct_devices'], d['state_identical_across_ranks'], ('%.1f MFFT/s' % (d['strong']['value']/1e6)) if 'strong' in d else '-')
    print(line)
PY
python -c "
import json; d=json.load(open('profiles/pmc_traffic.json'))
for k,v in d['entries'].items(): print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ('csrc_sha256','kernels','source')})"
