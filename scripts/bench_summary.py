import json,sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d['decode']['gzip_writer']; b = d['decode']['assembly']['bgzf']
print('ingest_full', round(d.get('ingest_full',{}).get('ms',0),1)); print('step', round(d['ms_per_step'], 2), 'K1', round(d['roofline']['ms_per_launch'],2), 'encode_full', round(d['encode_full']['ms'], 2), 'decode_full', round(d['decode_full']['ms'], 2), 'decode', round(d['decode']['ms_per_step'],2))
print('series: %.2f ms %d B (zlib5 %d), cut8k %.2f ms %d B; bgzf %.1f ms %d B' % (g['ms'], g['gz_bytes'], g['zlib_level5_bytes'], g['ms_cut_8k'], g['gz_bytes_cut_8k'], b['ms'], b['bgzf_bytes']))
