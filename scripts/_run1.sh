R=/root/repo; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ss; rocprofv3 --kernel-trace --output-format csv -d /tmp/ss -- python3 $R/scripts/_single_shot.py 2>/tmp/ss.err | tail -8
python3 $R/scripts/_show_shots.py /tmp/ss
