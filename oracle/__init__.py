"""oracle/ -- TEST INFRASTRUCTURE ONLY (checker + reported CPU baseline).

Nothing under pysurfinv_amd/ may import this package.
"""
