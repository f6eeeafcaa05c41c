"""``.mell`` container I/O: pickle (optionally gzip) of a dict, as written by the reference's generate_mel.py.

Mirrors reference MBExWN_NVoc/fileio/iovar.py:34-106 (save_var / load_var); dill is optional there and
is not needed for the plain dicts the CLI exchanges.
"""
import gzip
import pickle


def save_var(filename, data, protocol=-1, allow_dill=False):
    opener = gzip.open if filename.endswith(".gz") else open
    with opener(filename, "wb") as output:
        pickle.dump(data, output, protocol)


def load_var(filename):
    opener = gzip.open if filename.endswith(".gz") else open
    try:
        with opener(filename, "rb") as inp:
            return pickle.load(inp)
    except UnicodeDecodeError:
        # file written under python 2 (reference iovar.py:91-96)
        with opener(filename, "rb") as inp:
            return pickle.load(inp, encoding="latin1")
