"""Timeline of the last iterations in a rocprofv3 kernel trace (…_kernel_trace.csv): per launch the queue, start offset and duration in us, so that what
overlaps what (the exchange kernel of the communicator's stream against K1's bulk on the compute stream) can be read off.
usage: python tools/r3_timeline.py <kernel_trace.csv> [last_n_launches]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"]
    name = name[:70]
    print(f"q{q:>3} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  end {(e - t0) / 1e3:9.1f}  grid {r.get('Grid_Size_X', '?'):>8}  {name}")
