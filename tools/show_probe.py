import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else '/root/repo/gpurun_out/probe_tile.json'))
print(d['gemm_err_vs_f64_max_rms'])
for r in d['layers']:
    print("%-8s rows %6d %3d->%3d pairs %7d | gather f32 %6.0f x6 %6.0f x8 %6.0f | tile8 %6.0f/%6.0f tile6 %6.0f/%6.0f | plan %4.0f | frac8 %.3f" % (
        r['kind'], r['rows'], r['cin'], r['cout'], r['pairs'], r['gather_f32_us'], r['gather_bf16x6_us'], r['gather_bf16x8_us'], r['tile_8_map0_us'],
        r['tile_8_map1_us'], r['tile_6_map0_us'], r['tile_6_map1_us'], r['plan_us'], r['tile_8_frac_of_8TBs']))
