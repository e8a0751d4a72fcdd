# Which of the context's streams should share a hardware queue?  (JXLT_STREAM_ROLES)  Streams are made in
# the order main | copy, DC copy, upload, auxiliary, DC packing = queues 0 | 1 2 3 0 1; the digits say which of the five
# plays the role copy / DC copy / upload / auxiliary / DC packing.
for rep in 1 2; do
for roles in 01234 01324 03214 31204 04231 02134 34201 21034; do
  echo -n "[$roles] "; JXLT_STREAM_ROLES=$roles timeout 300 python tools/run_resident.py 16384 40 2>&1 | grep done | cut -c24-45 | tr '\n' ' '
  JXLT_STREAM_ROLES=$roles timeout 300 python tools/run_resident.py 4096 100 2>&1 | grep done | cut -c22-42 | tr '\n' ' '
  JXLT_STREAM_ROLES=$roles timeout 300 python tools/run_resident.py 2048 100 2>&1 | grep done | cut -c20-40
done
done
