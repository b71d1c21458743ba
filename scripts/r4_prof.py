"""decode the AZMI_PIPE_PROF lines of a pipe_bench log: per-pass averages between the last two prints"""
import sys
rows = [list(map(int, l.split()[2:])) for l in open(sys.argv[1]) if l.startswith("pipe prof:")]
a, b = rows[-2], rows[-1]
d = [y - x for x, y in zip(a, b)]
passes = max(1, d[2])
print("passes %.2fM  groups/pass %.2f  polls/pass %.1f" % (passes / 1e6, d[3] / passes, d[4] / passes))
print("per pass (us): sim loop %.1f  idle/token wait %.1f  io (load/store/tokens) %.1f  move step %.2f" % (d[0] / passes / 100, d[1] / passes / 100, d[5] / passes / 100, (d[15] if len(d) > 15 else 0) / passes / 100))
it = max(1, d[13])
print("group 0 per simulation (us): backup %.2f descent %.2f expansion %.2f probe+request %.2f ; sims/pass (group 0) %.2f ; levels/sim %.2f" % (
    d[9] / it / 100, d[10] / it / 100, d[11] / it / 100, d[12] / it / 100, it / passes, d[14] / it))
life = d[6]
print("wavefront lifetime share: sim %.2f idle %.2f io %.2f" % (d[0] / life, d[1] / life, d[5] / life))
print("net: waiting %.2f tile %.2f" % (d[7] / max(1, d[7] + d[8]), d[8] / max(1, d[7] + d[8])))
