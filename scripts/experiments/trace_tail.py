import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: the final 40 kernels
for r in rows[-int(sys.argv[2]):]:
    print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  {r["Kernel_Name"][:70]}')
