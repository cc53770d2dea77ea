"""Enum registries keyed by the reference CLI's string names.

Mirrors the behaviour of BaseEnumOptions / ClassEnumOptions / MethodMap in the reference
(/root/reference/utils/utils.py:297-315): a member's `.cls` is the registered class and calling
the member instantiates it; `list_names()` feeds the CLI choices.
"""
from collections import namedtuple
from enum import Flag, auto
from functools import partial


class BaseEnumOptions(Flag):
    def __str__(self):
        return self.name

    @classmethod
    def list_names(cls):
        return [member.name for member in cls]


class ClassEnumOptions(BaseEnumOptions):
    """Members are MethodMap(value, cls) pairs: Flag keeps `.value`, we expose `.cls`."""

    @property
    def cls(self):
        return self.value.cls

    def __call__(self, *args, **kwargs):
        return self.value.cls(*args, **kwargs)


# MethodMap(SomeClass) -> (auto(), SomeClass); Enum unpacks the first field as the flag value
MethodMap = partial(namedtuple("MethodMap", ["value", "cls"]), auto())
