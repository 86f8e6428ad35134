"""Durations of the fill and thread-per-read traceback launches of a `rocprofv3 --kernel-trace --output-format csv` run, in launch order.
  python tools/launch_times.py <rocprof output directory>"""
import sys, os, csv, glob
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
ks=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:24]) for r in rows if 'viterbi' in r['Kernel_Name']]
ks.sort()
fills=[round((e-s)/1e6,1) for s,e,n in ks if 'fill' in n]
tbs=[round((e-s)/1e6,1) for s,e,n in ks if 'traceback_kernel' in n]
print("fill", fills)
print("tb  ", tbs)
