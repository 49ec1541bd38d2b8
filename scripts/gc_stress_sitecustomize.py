"""Put a copy of this file as `sitecustomize.py` on PYTHONPATH to run the test-suite with the cyclic collector firing
~7x as often (and the oldest generation every few hundred allocations): a collection then falls into nearly every
HIP-graph capture, which is what killed the driver's round-2 GPU run (rc 134) on whatever allocation count its own
start-up hook happened to add.  scripts usage:
    mkdir /tmp/gcs && cp scripts/gc_stress_sitecustomize.py /tmp/gcs/sitecustomize.py
    PYTHONPATH=/tmp/gcs python -m pytest tests -m gpu -q"""
import gc

gc.set_threshold(100, 3, 3)
