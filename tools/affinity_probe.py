"""Where the host threads of a fit may run (development aid): the affinity of every thread of the process once a
HostPipeline is up."""
import os, sys, glob
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from fokl_gpy_amd import host_pipeline as hp, _capi
np.random.seed(1)
stream = _capi.LegacyStream()
pipe = hp.HostPipeline(stream, 2000)
masks = {}
for status in glob.glob('/proc/self/task/*/status'):
    text = open(status).read()
    name = text.split('\n')[0].split('\t')[1]
    cpus = [line.split('\t')[1] for line in text.split('\n') if line.startswith('Cpus_allowed_list')][0]
    masks.setdefault(cpus, []).append(name)
for cpus, names in masks.items():
    print(cpus, len(names), sorted(set(names)))
print('thread plan', hp._thread_plan(), 'bulk', hp._bulk_threads())
pipe.close()
