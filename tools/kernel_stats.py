# kernel_stats.py DIR -- the per-kernel table (calls, average us, share) of a rocprofv3 --kernel-trace --stats run under DIR
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1000)).rjust(10), r['Percentage'])
