class Error(Exception):
    pass


class UnsupportedMode(Error):
    pass


class DependencyNotInstalled(Error):
    pass
