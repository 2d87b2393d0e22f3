"""The reference's ``Runtime`` plugin base (utils/keras/runtimes/runtime.py:19-42): a
constructor that caches one engine per path in the class, ``__call__`` and ``load_engine``."""
from abc import ABCMeta, abstractmethod


class Runtime(metaclass=ABCMeta):
    _engines = {}

    def __init__(self, path, *, engine=None, reload=False, **kwargs):
        if engine is None:
            if path not in self._engines or reload:
                self._engines[path] = self.load_engine(path, **kwargs)
            engine = self._engines[path]
        self.path = path
        self.engine = engine

    def __repr__(self):
        return '<{} path={}>'.format(self.__class__.__name__, self.path)

    @abstractmethod
    def __call__(self, *args, **kwargs):
        """ Performs custom runtime inference """

    @staticmethod
    @abstractmethod
    def load_engine(path, **kwargs):
        """ Loads the custom runtime engine """
