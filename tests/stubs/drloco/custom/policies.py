"""Interface stub (see tests/stubs/README.md): the name an archive's `policy_class` entry resolves to."""


class CustomActorCriticPolicy:
    pass
