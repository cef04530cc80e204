"""Host-side mirrors of the reference's data path (lib/data_utils/data_workers.py) over the HIP kernels."""
