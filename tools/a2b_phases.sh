export HPE_EXTRA_FLAGS="-DHPE_A2B_STAMPS"
python tools/a2b_phases.py 2>&1 | grep "bounded"
unset HPE_EXTRA_FLAGS; python -c "from hpe_amd import build; build.build()"
