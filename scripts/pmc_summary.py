"""Build a profiles/rNN_pmc_summary_*.json from the rocprofv3 --pmc passes of scripts/pmc_layer.sh.
    python scripts/pmc_summary.py gpurun_out/pmc_<tag> profiles/r01_pmc_summary_f16x3.json [batch_per_gpu] [bench args] [kernel_stats.csv] [calls]
Kernels are keyed by their base name; variants that do different work per launch keep their own key
("iaf_layer_h_kernel<first>": start conv fused in; "iaf_layer_c_kernel<head>": flow head in the epilogue; "iaf_group_kernel<first>" / "<head>": layer groups that open / close a flow;
"iaf_pair_c_kernel<...>": per template arguments)."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd import build as wbuild

src, dst = sys.argv[1], sys.argv[2]
try:
    with open(os.path.join(src, 'source_hash.txt')) as f:
        measured_hash = f.read().strip()
except OSError:
    sys.exit('{}: no source_hash.txt (scripts/pmc_layer.sh writes it when the counters are collected); refusing to '
             'stamp counters of unknown sources'.format(src))
if len(measured_hash) != 64:
    sys.exit('{}/source_hash.txt does not hold a sha256'.format(src))
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
extra = sys.argv[4] if len(sys.argv) > 4 else ''
stats_csv = sys.argv[5] if len(sys.argv) > 5 else None      # rocprofv3 --kernel-trace --stats of the same command, same run
stats_steps = int(sys.argv[6]) if len(sys.argv) > 6 else 23  # calls of the generate path in that run (steps + warm-up)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()


def key_of(name):
    m = re.search(r'(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)(<[^>]*>)?\(', name)
    if not m:
        return None
    base, targs = m.group(1), m.group(2) or ''
    if base == 'iaf_layer_h_kernel' and 'true' in targs:
        return base + '<first>'
    if base == 'iaf_layer_c_kernel':      # template <HN, LAST, W2, DMA, NOPF>: only LAST changes the work per launch
        a = [x.strip() for x in targs.strip('<>').split(',')]
        return base + '<head>' if len(a) > 1 and a[1] == 'true' else base
    if base == 'iaf_group_kernel':       # template <FIRST, LAST>: the start conv instead of reading l / the head instead of writing l
        a = [x.strip() for x in targs.strip('<>').split(',')]
        tag = ','.join(t for t, on in (('first', a[0] == 'true'), ('head', len(a) > 1 and a[1] == 'true')) if on)
        return base + ('<' + tag + '>' if tag else '')
    if base == 'iaf_pair_c_kernel':
        return base + targs.replace(' ', '')
    if base in ('deconv_mfma_h_kernel', 'deconv_mfma_hs_kernel'):
        return base + targs.replace(' ', '')
    if base == 'iaf_layer_kernel':       # fp32 form, template <HOIST, LAST, START>: hoisted conditioning / the flow head in the epilogue / the start conv in front
        a = [x.strip() for x in targs.strip('<>').split(',')]
        tag = ','.join(t for t, on in (('hoist', a[0] == 'true'), ('head', len(a) > 1 and a[1] == 'true'),
                                       ('start', len(a) > 2 and a[2] == 'true')) if on)
        return base + ('<' + tag + '>' if tag else '')
    if base == 'gemm_f32_kernel':        # template <NB, DECONV>: the conditioning GEMM / the upsampler's last layer
        a = [x.strip() for x in targs.strip('<>').split(',')]
        return base + ('<deconv>' if len(a) > 1 and a[1] == 'true' else '<cond>')
    return base


for sub in ('sq', 'fetch', 'write', 'inst'):
    for f in glob.glob(os.path.join(src, sub, '*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            k = key_of(row['Kernel_Name'])
            if k is None or not (k.startswith('iaf_') or k.startswith('deconv_') or k.startswith('gemm_f32')):
                continue
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            cnt[(k, row['Counter_Name'])] += 1
kernels = {}
for k, d in agg.items():
    e = {c: round(v / cnt[(k, c)], 1) for c, v in d.items()}
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        e['hbm_bytes_per_launch'] = int((2 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024)
    if e.get('SQ_VALU_MFMA_BUSY_CYCLES') and e.get('GRBM_GUI_ACTIVE'):
        # matrix-pipe busy cycles summed over the chip's 1024 SIMDs / (1024 x cycles of the launch); GRBM_GUI_ACTIVE
        # is summed over the 8 XCDs (scripts/ubench/run_issue_overlap.sh calibrates both on hand-placed MFMA streams)
        e['mfma_util'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (e['GRBM_GUI_ACTIVE'] / 8.0), 4)
    kernels[k] = e
kernel_us = {}
if stats_csv:
    tot = collections.defaultdict(float)
    for row in csv.DictReader(open(stats_csv)):
        k = key_of(row['Name'])
        if k is None:
            continue
        tot[k] += float(row['TotalDurationNs'])
    kernel_us = {k: round(v / 1e3 / stats_steps, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1]) if v / 1e3 / stats_steps >= 0.5}
out = {
    '_about': 'rocprofv3 --pmc passes (scripts/pmc_layer.sh: SQ pass, FETCH_SIZE pass, WRITE_SIZE pass, instruction-mix '
              'pass; kernel-trace only) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline ' + extra + '`, one MI355X, '
              + ('fp32-MFMA' if 'f32' in extra else 'split-fp16 (f16x3)') + ' path. Per-dispatch averages; FETCH_SIZE/WRITE_SIZE in KiB; '
              'hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports half of a coalesced '
              'stream, MI355X_MICROARCH.md; calibrated on iaf_head_h_kernel: it reads 98.3 MB exactly once and reports '
              '~48.8 MB). These fabric-side counters include Infinity Cache hits.',
    'workload': {'batch_per_gpu': batch, 'frames': 384, 'samples': 76800},
    # hash of the kernel sources these counters were measured on (nsynth_wavenet_amd.build.source_hash, written by
    # scripts/pmc_layer.sh at collection time): bench.py replays `hbm_bytes_per_launch` as roofline.traffic only
    # while the sources it runs are the same
    'source_hash': measured_hash,
    'kernels': kernels,
    # microseconds per generate call and kernel (rocprofv3 --kernel-trace --stats of the same command in the same run;
    # kernels under 0.5 us per call left out): where the call's time goes
    'kernel_us_per_call': kernel_us,
}
json.dump(out, open(dst, 'w'), indent=1)
for k in sorted(kernels):
    print(k, kernels[k].get('hbm_bytes_per_launch'))
