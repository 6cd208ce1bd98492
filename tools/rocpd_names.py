import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
for r in db.execute("select distinct kernel_name from counters_collection"): print(r[0][:150])
