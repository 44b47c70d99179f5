"""Import stub (plotting only in the reference: utils/data_utils.py:8). Used only by gen_goldens.py."""
