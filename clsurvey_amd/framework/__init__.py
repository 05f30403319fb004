"""Thin Continual-Hyperparameter-Framework driver (counterpart of src/framework/)."""
