#!/bin/bash
# gpurun_out/<tag>/ (tools/final_round.sh on the GPU box) -> the committed profiles/<round>_* files. Usage: tools/refresh_profiles.sh <tag> <round>
T=$1; R=${2:-r03}
python tools/write_profiles.py $T $R | tail -1
python tools/write_sq_profile.py ${T}_sq $R 1200000000 | tail -1
[ -d gpurun_out/${T}_sq4 ] && python tools/write_sq_profile.py ${T}_sq4 ${R}_cfg4 1200000000 | tail -1
python tools/write_sharded_profile.py $T $R > /dev/null
for c in cfg3 cfg4 merge dup; do tail -1 gpurun_out/$T/bench_$c.json > profiles/${R}_bench_$c.json; done
tail -1 gpurun_out/$T/bench_line.json > profiles/${R}_bench_cfg2.json
python - "$T" "$R" <<'PY'
import json, sys
T, R = sys.argv[1], sys.argv[2]
m = {'cfg 2 weak scaling: 8 x 10 M x 150 bp, K=31, PB=24': 'emul_reads10000000.json', 'cfg 5 operand: 8 x 6.25 M x 150 bp, K=31, PB=24': 'emul_cfg5.json',
     'cfg 3: 8 x 12.5 M x 150 bp, K=31, PB=28': 'emul_reads12500000prefixbits2.json', 'cfg 4: 8 x 6.25 M x 250 bp, K=59, PB=28': 'emul_k59prefixbits28reads6250.json'}
out = {}
for k, f in m.items():
    out[k] = json.loads(open(f'gpurun_out/{T}/' + f).read().strip().splitlines()[-1])
    b = out[k]['build']
    print(k[:6], round(b['receiver_ms'], 1), {a: round(x, 1) for a, x in b['receiver_stage_ms'].items() if a.startswith('bucket')}, (out[k].get('merge') or {}).get('ms'))
json.dump(out, open(f'profiles/{R}_emulated_rank.json', 'w'))
d = json.loads(open(f'profiles/{R}_bench_cfg2.json').read())
print('cfg2', d['ms_per_step'], d['value'], 'h2d', d['h2d_inclusive']['ms_per_step'], 'fasta', d['fasta_inclusive']['ms_per_step'], 'per record', d['per_record']['ms_total'],
      'serialize', d['serialize']['to_host_ms'], 'frac', d['roofline']['frac'])
PY
W=""; for f in wire_cfg3 wire_cfg3_rank3 wire_cfg3_rank6 wire_cfg3_rank7 wire_cfg2 wire_cfg4 wire_cfg4_rank7 wire_w2_replicate wire_w2_sorted wire_w2_bins wire_w3_replicate wire_w3_sorted wire_w3_bins wire_w4_replicate wire_w4_sorted wire_w4_bins wire_w5_replicate wire_w5_sorted wire_w5_bins wire_w2_replicate_cfg4; do [ -s gpurun_out/$T/$f.json ] && W="$W gpurun_out/$T/$f.json"; done
python tools/write_wire_profile.py profiles/${R}_wire_emulated.md $R $W
[ -s gpurun_out/$T/bench_cpufull.json ] && tail -1 gpurun_out/$T/bench_cpufull.json | python -c "import json,sys; json.dump(json.loads(sys.stdin.read()), open('profiles/${R}_bench_cpufull.json','w'), indent=1)"
for c in cfg3 cfg4 merge; do
  [ -s gpurun_out/$T/bench_cpufull_$c.json ] && tail -1 gpurun_out/$T/bench_cpufull_$c.json | python -c "import json,sys; json.dump(json.loads(sys.stdin.read()), open('profiles/${R}_bench_cpufull_$c.json','w'), indent=1)"
  [ -s gpurun_out/$T/kernel_stats_$c.md ] && { echo "# ${R} — rocprofv3 --kernel-trace --stats, \`bench.py --config $c --steps 3 --warmup 1\` (1 x MI355X; tools/r5_final.sh kstats, summarised by tools/rocpd_summary.py; the table also holds the input generation and, for the merge, the two index builds in front of the timed steps)"; echo; cat gpurun_out/$T/kernel_stats_$c.md; } > profiles/${R}_kernel_stats_$c.md
done
python tools/write_emulated_rank_md.py $R r04
# round 6: the bucket kernels alone (tools/dev_msd_bench.cpp) and their SQ counters
if [ -s gpurun_out/$T/msd_bench_cfg2.txt ]; then
  { echo "# ${R} — the bucket kernels alone on the real buckets of a cfg-2 build (tools/dev_msd_bench.cpp: K = 31, PREFIX_BITS = 24, 10 M x 150 bp; best / average of 5 launches over pristine copies of the arena;"
    echo "\`class\` rows = k_bucket_msd, \`sorted\` rows = k_bucket_sorted on the same class list; \`chk\` = checksum of counts, kinds and words of the class: equal = same result)"; echo "(the harness keeps the five classes of rounds 1 - 5; in the library the 129 - 256-word runs of class 1 now take \`<64, 256>\`: profiles/${R}_msd_bench_split.md)"; echo; echo '```'; cat gpurun_out/$T/msd_bench_cfg2.txt; echo '```'; } > profiles/${R}_msd_bench.md
fi
if [ -d gpurun_out/${T}_msdsq ]; then
  { echo "# ${R} — SQ counters of the bucket kernels alone (tools/r6_msd_counters.sh: rocprofv3 --kernel-trace --pmc, one pass per counter group, tools/dev_msd_bench.bin directly behind \`--\`; per dispatch, summed over the chip)"; echo
    cat gpurun_out/${T}_msdsq/sq1.md; echo; cat gpurun_out/${T}_msdsq/sq2.md; echo; cat gpurun_out/${T}_msdsq/sq3.md; } > profiles/${R}_msd_sq_counters.md
fi
