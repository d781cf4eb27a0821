"""Interface stub (see tests/stubs/README.md): only the class PATH drloco.custom.policies.CustomActorCriticPolicy, which an SB3 model.zip of the
reference names in its `data` member (drloco/custom/policies.py:55-80 is the real class)."""
