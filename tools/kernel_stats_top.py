import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms',tot/1e6)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    print(r['Name'][:100], r['Calls'], '%.1f us avg'%(float(r['AverageNs'])/1e3), '%.1f ms'%(float(r['TotalDurationNs'])/1e6))
