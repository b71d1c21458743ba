import json,sys
j=json.load(open(sys.argv[1])); c=j["config"]
print(sys.argv[2] if len(sys.argv)>2 else "", "games/s %.0f" % j["value"], "ms/round %.4f" % c["ms_per_round"], "tree %.1f us net %.1f us" % (1e3*c["tree_kernel_ms"], 1e3*c["net_ms"]), "hit %.3f" % c["cache_hit_rate"], "sims/s %.1fM evals/s %.1fM" % (c["sims_per_s"]/1e6, c["leaf_evals_per_s"]/1e6), "frac %.3f agg %.3f" % (j["roofline"]["frac"], j["roofline"]["aggregate_frac"]))
