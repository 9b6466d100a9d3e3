# Bulk-update groups of the factorisation: CBO_HIP_BULK_GROUP = 1 (pairs), 2, 4 (default) and the row threshold of the groups of
# four; digests of the factor (same bits or not) and the fit time.  usage: bash scripts/probes/bulk_group_scan.sh
for g in 1 2 4; do
  echo "== CBO_HIP_BULK_GROUP=$g"; CBO_HIP_BULK_GROUP=$g python scripts/factor_digest.py 8192 12288 16384; CBO_HIP_BULK_GROUP=$g python scripts/chol_timing.py | grep "n=8192\|n=16384"
done
for r in 4096 6144 10240; do
  echo "== CBO_HIP_BULK_GROUP=4 CBO_HIP_BULK_GROUP4_ROWS=$r"; CBO_HIP_BULK_GROUP4_ROWS=$r python scripts/factor_digest.py 16384; CBO_HIP_BULK_GROUP4_ROWS=$r python scripts/chol_timing.py | grep "n=8192\|n=16384"
done
