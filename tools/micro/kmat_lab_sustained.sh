# sustained (about a second each) timings of chosen variants of tools/micro/kmat_lab: usage kmat_lab_sustained.sh N kind reps name...
L=tools/micro/_bin/kmat_lab
N=$1; K=$2; R=$3; shift 3
for v in "$@"; do LAB_ONLY="$v" timeout -k 5 60 $L $N $K 10 $R | grep -v differing; done
