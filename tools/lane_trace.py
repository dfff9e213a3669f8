import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
seq=[(r['Kernel_Name'][:40], int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Grid_Size_X']), r['Queue_Id'], r['Stream_Id']) for r in rows]
seq.sort(key=lambda t:t[1])
t0=seq[len(seq)//2][1]
for t in seq[len(seq)//2: len(seq)//2+40]:
    print(f"{t[0]:42s} start={(t[1]-t0)/1e3:9.1f} end={(t[2]-t0)/1e3:9.1f} grid={t[3]:8d} q={t[4]} s={t[5]}")
