import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
c=collections.Counter((r['Queue_Id'],r['Stream_Id']) for r in rows)
for k,v in sorted(c.items()): print(k,v)
