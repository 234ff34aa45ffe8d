"""Does a timing event recorded inside a captured hipGraph (torch `external` events) read back after a replay?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
print("external events work:", bench.external_events_work(dev))
x = torch.ones(10, device=dev)
print("runtime alive afterwards:", float((x + 1).sum()))
