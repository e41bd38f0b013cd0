"""Packaging of the host side.  The HIP library is built in-tree (`make -C snout_amd/csrc`, or
`python -c 'import __graft_entry__ as g; g.build()'`) and shipped as package data; console entries mirror
the reference's (`snout = snout.cli:main`, /root/reference/setup.py:52-54) plus the `btle_rx` name its BTLE
scan resolves on $PATH (snout/util/btle.py:53)."""
from setuptools import setup

setup(
    name="snout_amd",
    version="0.2.0",
    description="MI355X-native IQ->packets receive path behind Snout's scan interface",
    packages=["snout_amd"],
    package_data={"snout_amd": ["lib/libsnout_rx.so"]},
    include_package_data=True,
    install_requires=["numpy", "click"],
    entry_points={"console_scripts": ["snout = snout_amd.cli:main", "btle_rx = snout_amd.cli:btle_rx_main"]},
)
