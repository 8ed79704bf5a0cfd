from .multisample import obs_to_sample

__all__ = ['obs_to_sample']
