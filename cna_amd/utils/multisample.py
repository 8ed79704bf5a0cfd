"""Sample-level metadata from per-cell annotations (reference utils/multisample.py:4-11).
Host-side pandas helper; callers use it to build ``y`` / ``covs`` / ``batches`` before
``association`` (demo/demo.ipynb:115)."""
import pandas as pd


def obs_to_sample(d, columns, sid_name, aggregate='mean'):
    """One row per sample id (in order of first appearance), ``columns`` aggregated over
    that sample's cells."""
    if isinstance(columns, str):
        columns = [columns]
    samplem = pd.DataFrame(index=d.obs[sid_name].unique())
    samplem[columns] = d.obs.groupby(by=sid_name)[columns].aggregate(aggregate)
    return samplem
